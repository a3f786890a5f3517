"""SF6: decoder::make accepts it (decoder_impl.cc:57).  An explicit header needs 5 codewords and an SF6 header block (4 + 4 symbols of SF - 2 = 4 bits) holds
4, so only the implicit header means anything there; both are held to the oracle here - frames, header positions and the complete work() trace.  At
decimation 8 / 4 SF6 runs walker2 on lora_wave_decim.inc.hip's demodulators (round 6: `walker2_kernel_sf6_d8 / _d4[_grad]`; by sample count its windows are
SF7's at decimation 4 / 2), at other decimations the generic kernels."""
import numpy as np
import pytest

from gr_lora_amd import synth

pytestmark = pytest.mark.gpu


def _dev(iq):
    import torch
    return torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).cuda()


@pytest.mark.parametrize("decim", [8, 4, 16])
@pytest.mark.parametrize("demod", [0, 1, 2])
@pytest.mark.parametrize("implicit", [True, False])
def test_sf6_equals_oracle(oracle_mod, demod, implicit, decim):
    import torch
    from gr_lora_amd import capi
    rate = 125000.0 * decim
    kw = dict(sf=6, cr=3, crc=False, implicit=implicit, samp_rate=rate)
    cfg = synth.TxConfig(sf=6, cr=3, crc=False, implicit=True, samp_rate=rate)   # (what is on air has no header either way: the explicit-header decoder reads one out of the first block)
    rng = np.random.default_rng(60 + demod + 2 * implicit)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(3, 30)), dtype=np.uint8)) for _ in range(8)]
    st = synth.build_stream(payloads, cfg, rng=rng, gap_symbols=(3.0, 9.0), noise_sigma=synth.awgn_sigma_for_snr(45.0, cfg))
    o = oracle_mod.Oracle(demod=demod, **kw)
    o.enable_trace()
    o.run(st.iq)
    h = capi.Handle(demod=demod, flags=capi.FLAG_TRACE, **kw)
    want = "walker_kernel" if decim == 16 else "walker2_kernel_sf6_d%d%s" % (decim, "_grad" if demod == 0 else "")
    assert h.kernel_name().startswith(want), h.kernel_name()
    dev = _dev(st.iq)
    h.decode_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size], torch.cuda.current_stream().cuda_stream)
    got, tr = h.drain(), h.trace()
    h.close()
    assert [g for g, _ in got] == o.frames()
    assert [i.header_pos for _, i in got] == o.frame_positions()
    otr = o.trace()
    assert len(tr) == len(otr)
    bad = [(a, b) for a, b in zip(tr, otr) if (a[0], a[1], a[2], a[3], a[4]) != (b[0], b[1], b[2], b[3], b[4])]
    assert not bad, bad[:4]
    if implicit:
        assert len(got) >= 6


@pytest.mark.parametrize("decim", [8, 4])
def test_sf6_symbol_kernels_vs_oracle(oracle_mod, decim):
    """the symbol-level kernels (lora_hip_demod_symbols_ex_device) at SF6: get_shift_fft / max_frequency_gradient_idx + fine_sync window by window"""
    import test_gpu_decim as T
    T.RATES.setdefault(8, 1e6)
    T.test_fft_shift_and_fine_sync_vs_oracle(oracle_mod, 6, decim)
    T.test_gradient_bin_and_fine_sync_vs_oracle(oracle_mod, 6, decim)
