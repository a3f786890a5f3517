"""the two constructor switches of decoder::make that change the receive path -
implicit header (no header parse, payload ends when the symbol energy halves, decoder_impl.cc:828-829,:861-864) and
disable_drift_correction (d_enable_fine_sync = false, :90,:501) - on every walker kernel (walker2 SF7/8, walker3
SF9-12, the generic kernel for the gradient demodulator and for implicit mode at SF7/8), against the CPU oracle:
frames, header positions and the complete work() trace."""
import numpy as np
import pytest

from gr_lora_amd import synth

pytestmark = pytest.mark.gpu


def _dev(iq):
    import torch
    return torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).cuda()


def _compare(oracle_mod, iq, demod, exact=True, **kw):
    from gr_lora_amd import capi
    from parity_util import assert_trace_parity
    o = oracle_mod.Oracle(demod=demod, **kw)
    o.enable_trace()
    o.run(iq)
    h = capi.Handle(demod=demod, flags=capi.FLAG_TRACE, **kw)
    dev = _dev(iq)
    h.decode_device(dev.data_ptr(), iq.size, [0], [iq.size], 0)
    got = h.drain()
    tr = h.trace()
    h.close()
    assert [g.hex() for g, _ in got] == [f.hex() for f in o.frames()], (demod, kw)
    if exact:
        assert [i.header_pos for _, i in got] == o.frame_positions(), (demod, kw)
    else:
        assert all(abs(i.header_pos - p) <= 1 for (_, i), p in zip(got, o.frame_positions())), (demod, kw)
    assert_trace_parity(tr, o.trace(), exact, (demod, kw))   # (no window is exempt: samples of exactly zero follow the reference's std::arg(0), tests/test_gpu_zeros.py)
    return len(got)


@pytest.mark.parametrize("sf", [7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("demod", [0, 1, 2])
def test_disable_drift_correction(oracle_mod, sf, demod):
    n = {7: 6, 8: 5, 9: 4, 10: 3, 11: 2, 12: 1}[sf]
    for cr in ((4, 1) if sf < 11 else (4,)):
        cfg = synth.TxConfig(sf=sf, cr=cr, reduced_rate=(sf > 10))
        rng = np.random.default_rng(31 * sf + cr)
        payloads = [bytes(rng.integers(0, 256, int(rng.integers(4, 28)), dtype=np.uint8)) for _ in range(n)]
        st = synth.build_stream(payloads, cfg, rng=rng, noise_sigma=synth.awgn_sigma_for_snr(42.0, cfg))
        got = _compare(oracle_mod, st.iq, demod, sf=sf, cr=cr, reduced_rate=(sf > 10), disable_drift_correction=True)
        assert got == n
        if True:
            st = synth.build_stream(payloads, cfg, rng=np.random.default_rng(5 * sf + cr))
            assert _compare(oracle_mod, st.iq, demod, exact=True, sf=sf, cr=cr, reduced_rate=(sf > 10), disable_drift_correction=True) == n


@pytest.mark.parametrize("sf", [7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("demod", [0, 1, 2])
def test_implicit_header(oracle_mod, sf, demod):
    """implicit header: cr / crc come from the constructor, the payload ends by energy."""
    n = {7: 5, 8: 4, 9: 3, 10: 3, 11: 2, 12: 1}[sf]
    for cr, crc in (((4, True), (2, False)) if sf < 11 else ((3, True),)):
        cfg = synth.TxConfig(sf=sf, cr=cr, crc=crc, reduced_rate=(sf > 10), implicit=True)
        rng = np.random.default_rng(17 * sf + cr)
        payloads = [bytes(rng.integers(0, 256, int(rng.integers(6, 24)), dtype=np.uint8)) for _ in range(n)]
        st = synth.build_stream(payloads, cfg, rng=rng)
        got = _compare(oracle_mod, st.iq, demod, sf=sf, cr=cr, crc=crc, reduced_rate=(sf > 10), implicit=True)
        assert got >= 1


@pytest.mark.parametrize("sf", [9, 10, 11, 12])
def test_walker3_noisy_mixed_cr_segments(oracle_mod, sf):
    """walker3 across segment cuts: several packets of mixed CR with AWGN, small forced segments; frames and positions
    identical to the serial oracle (FFT_COMPAT and FFT)."""
    from gr_lora_amd import capi
    rng = np.random.default_rng(900 + sf)
    n = {9: 8, 10: 6, 11: 4, 12: 3}[sf]
    pieces = []
    for i in range(n):
        cfg = synth.TxConfig(sf=sf, cr=int(rng.integers(1, 5)), reduced_rate=(sf > 10))
        p = bytes(rng.integers(0, 256, int(rng.integers(3, 20)), dtype=np.uint8))
        pieces.append(synth.build_stream([p], cfg, rng=rng, tail_symbols=0.0).iq)
    sps = 8 << sf
    iq = np.concatenate(pieces + [np.zeros(3 * sps, np.complex64)])
    sigma = synth.awgn_sigma_for_snr(40.0, synth.TxConfig(sf=sf))
    iq = (iq + (rng.standard_normal(iq.size) + 1j * rng.standard_normal(iq.size)).astype(np.complex64) * np.float32(sigma / np.sqrt(2))).astype(np.complex64)
    dev = _dev(iq)
    for demod in (2, 1):
        o = oracle_mod.Oracle(sf=sf, cr=4, reduced_rate=(sf > 10), demod=demod)
        o.run(iq)
        for seg in (0, 24, 61):
            h = capi.Handle(sf=sf, cr=4, reduced_rate=(sf > 10), demod=demod, segment_symbols=seg)
            h.decode_device(dev.data_ptr(), iq.size, [0], [iq.size], 0)
            got = h.drain()
            h.close()
            assert [g.hex() for g, _ in got] == [f.hex() for f in o.frames()], (sf, demod, seg)
            assert [i.header_pos for _, i in got] == o.frame_positions(), (sf, demod, seg)
