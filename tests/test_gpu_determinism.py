"""Stress: the same batch decoded repeatedly must give identical, correct frames every time,
with enough independent jobs that several workgroups share every CU (this caught a
register-reuse-before-store-drained bug in the 512-thread walker)."""
import numpy as np
import pytest

from gr_lora_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sf,n_streams", [(7, 1024), (8, 512)])
def test_repeated_decode_is_deterministic_and_correct(sf, n_streams):
    import torch
    assert torch.cuda.is_available()
    from gr_lora_amd import capi
    cfg = synth.TxConfig(sf=sf, cr=4)
    rng = np.random.default_rng(31 + sf)
    base = [bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(16)]
    streams, offs, lens, expect = [], [], [], []
    off = 0
    for s in range(n_streams):
        p = base[s % 16]
        st = synth.build_stream([p], cfg, gaps=[int(rng.integers(2, 6) * cfg.sps)], tail_symbols=2.5)
        streams.append(st.iq); offs.append(off); lens.append(st.iq.size); off += st.iq.size
        expect.append(synth.expected_frame_tail(p, cfg))
    iq = np.concatenate(streams)
    dev = torch.from_numpy(iq.view(np.float32)).cuda()
    h = capi.Handle(sf=sf, cr=4, demod=capi.DEMOD_FFT_COMPAT)
    for it in range(25):
        h.decode_device(dev.data_ptr(), iq.size, offs, lens, 0)
        got = {}
        for b, i in h.drain():
            got.setdefault(i.stream, []).append(b[15:])
        wrong = [s for s in range(n_streams) if got.get(s, []) != [expect[s]]]
        assert not wrong, (it, wrong[:8])
    h.close()
