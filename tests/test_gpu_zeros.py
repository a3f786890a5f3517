"""std::arg(0) = 0: instantaneous_frequency (decoder_impl.cc:224-244) takes the phase of every SAMPLE, so next to a sample that is exactly
zero it returns -arg(x[k-1]) and +arg(x[k+1]); the kernels take the phase of the PRODUCT x[k+1] conj(x[k]), which is zero there.  The
kernels therefore watch for such products (they poison the sums they feed) and re-evaluate a window that holds one sample by sample, with
the reference's convention.  Here: streams with exact zeros planted inside the preamble, the SFD, the header and the payload - single
samples and short runs, as a saturating front end or a zero-stuffed capture produces them - on every kernel family and both estimators,
held to the oracle's complete work() trace (state, position, consume, bin, d_fine_sync, decision values), frames and header positions.
Tracing switches the decoupled passes off, so the payload-pass kernels (the symbol-level demodulators and their second reads, which decide
the published frames of every small pass) get their own cases without the trace: frames and header positions under LORA_HIP_DECOUPLED=1;
the generic kernels (decimation != 8, SF6, LORA_HIP_NO_FAST) theirs with the implicit header and with LORA_HIP_NO_FAST.
No window is exempt: tests/parity_util.py's windows_with_exact_zeros is gone."""
import numpy as np
import pytest

from gr_lora_amd import synth

pytestmark = pytest.mark.gpu


def _dev(iq):
    import torch
    return torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).cuda()


def _planted(sf, cr, seed, n_packets, density, runs):
    cfg = synth.TxConfig(sf=sf, cr=cr, reduced_rate=(sf > 10))
    rng = np.random.default_rng(seed)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(6, 20)), dtype=np.uint8)) for _ in range(n_packets)]
    st = synth.build_stream(payloads, cfg, rng=rng, noise_sigma=synth.awgn_sigma_for_snr(40.0, cfg))
    iq = st.iq.copy()
    n = iq.size
    for p in rng.integers(0, n, max(4, int(n * density))):      # single samples anywhere
        iq[p] = 0
    for p in rng.integers(0, n - 8, runs):                      # short runs
        iq[p:p + int(rng.integers(2, 6))] = 0
    iq[rng.integers(0, n, 3)] = np.complex64(complex(-0.0, 0.0))   # arg(-0 + 0j) = pi
    iq[rng.integers(0, n, 3)] = np.complex64(complex(0.0, -0.0))   # arg(+0 - 0j) = -0
    return iq


def _compare(oracle_mod, iq, demod, trace=True, **kw):
    from gr_lora_amd import capi
    from parity_util import assert_trace_parity
    o = oracle_mod.Oracle(demod=demod, **kw)
    o.enable_trace()
    o.run(iq)
    h = capi.Handle(demod=demod, flags=capi.FLAG_TRACE if trace else 0, **kw)
    dev = _dev(iq)
    h.decode_device(dev.data_ptr(), iq.size, [0], [iq.size], 0)
    got = h.drain()
    tr = h.trace() if trace else None
    h.close()
    if trace:
        assert_trace_parity(tr, o.trace(), True, (demod, kw))
    assert [g.hex() for g, _ in got] == [f.hex() for f in o.frames()], (demod, kw)
    assert [i.header_pos for _, i in got] == o.frame_positions(), (demod, kw)
    return sum(1 for s in o.trace() if s[0] in (4, 5)), len(o.frames())


@pytest.mark.parametrize("sf", [7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("demod", [0, 2])
def test_exact_zero_samples_follow_the_reference(oracle_mod, sf, demod):
    n_packets = {7: 5, 8: 4, 9: 3, 10: 2, 11: 2, 12: 1}[sf]
    decoded = frames = 0
    sps = 8 << sf
    for seed, density, runs in ((11 * sf + demod, 0.4 / sps, 6), (13 * sf + demod, 1.5 / sps, 0)):   # zeros per symbol window on average
        iq = _planted(sf, 4, seed, n_packets, density, runs)
        d, f = _compare(oracle_mod, iq, demod, sf=sf, cr=4, reduced_rate=(sf > 10))
        decoded += d
        frames += f
    assert decoded > 8           # the planted zeros leave the receiver working: header / payload windows were compared


@pytest.mark.parametrize("sf,demod", [(7, 0), (7, 2), (9, 0), (9, 2)])
def test_exact_zeros_without_drift_correction(oracle_mod, sf, demod):
    """d_enable_fine_sync = false: the gradient estimator then has no fine_sync sum to carry the poison"""
    iq = _planted(sf, 4, 400 + sf + demod, 3, 0.7 / (8 << sf), 4)
    _compare(oracle_mod, iq, demod, sf=sf, cr=4, disable_drift_correction=True)


@pytest.mark.parametrize("sf", [7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("demod", [0, 2])
def test_exact_zeros_in_the_payload_pass(oracle_mod, monkeypatch, sf, demod):
    """a decoupled pass (header-only walkers + the symbol-level kernels + payload_chain_kernel; never taken while tracing): the kPoisonBin / kFinePoison
    re-runs of demod_symbols_wave_kernel<7/8/9>, demod_symbols_wave_grad_kernel, demod_symbols_w3_kernel and demod_symbols_w3_grad_kernel and the
    second reads' poison skip decide these frames"""
    from gr_lora_amd import capi
    monkeypatch.setenv("LORA_HIP_DECOUPLED", "1")
    n_packets = {7: 5, 8: 4, 9: 3, 10: 2, 11: 2, 12: 1}[sf]
    sps = 8 << sf
    frames = 0
    for seed, density, runs in ((17 * sf + demod, 0.4 / sps, 6), (19 * sf + demod, 1.5 / sps, 0)):
        iq = _planted(sf, 4, seed, n_packets, density, runs)
        h = capi.Handle(demod=demod, sf=sf, cr=4, reduced_rate=(sf > 10))
        dev = _dev(iq)
        h.decode_device(dev.data_ptr(), iq.size, [0], [iq.size], 0)
        info = h.payload_pass()
        h.close()
        _, f = _compare(oracle_mod, iq, demod, trace=False, sf=sf, cr=4, reduced_rate=(sf > 10))
        frames += f
        assert info["packets"] > 0 or f == 0, info   # the pass was a decoupled one
    assert frames > 0


@pytest.mark.parametrize("sf,demod", [(7, 0), (7, 2), (9, 2)])
def test_exact_zeros_in_the_generic_kernels(oracle_mod, monkeypatch, sf, demod):
    """walker_kernel* (compute_ifreq's ifreq_prod_z_inl): what serves every configuration outside the fast families - here forced with LORA_HIP_NO_FAST,
    and with the implicit header (which the fast families decode as well, by energy)"""
    monkeypatch.setenv("LORA_HIP_NO_FAST", "1")
    iq = _planted(sf, 4, 500 + sf + demod, 3, 0.7 / (8 << sf), 4)
    _compare(oracle_mod, iq, demod, sf=sf, cr=4)
    monkeypatch.delenv("LORA_HIP_NO_FAST")
    cfg = synth.TxConfig(sf=sf, cr=4, implicit=True)
    rng = np.random.default_rng(600 + sf + demod)
    st = synth.build_stream([bytes(rng.integers(0, 256, 12, dtype=np.uint8)) for _ in range(3)], cfg, rng=rng, noise_sigma=synth.awgn_sigma_for_snr(40.0, cfg))
    iq = st.iq.copy()
    for p in rng.integers(0, iq.size, 12):
        iq[p] = 0
    _compare(oracle_mod, iq, demod, sf=sf, cr=4, implicit=True)
