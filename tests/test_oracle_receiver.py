"""Receiver-level behaviour of the oracle: the reference's own test matrices
(apps/generate_test_suites.py:157-203, scored like python/qa_testsuite.py:96-121
by whole-payload equality) on synthesised streams, plus the documented quirks."""
import numpy as np
import pytest

from gr_lora_amd import synth

SHORT_PAYLOADS = [("deadbeef", 5), ("88", 1), ("ffff", 10)]   # generate_test_suites.py:173-203


def _run(O, st, cfg, mode=0, ctor_cr=4, **kw):
    o = O.Oracle(sf=cfg.sf, cr=ctor_cr, crc=cfg.crc, reduced_rate=cfg.reduced_rate, implicit=cfg.implicit, demod=mode, **kw)
    o.run(st.iq)
    return o


@pytest.mark.parametrize("sf", [7, 8, 9, 10])
@pytest.mark.parametrize("cr", [1, 2, 3, 4])
def test_short_suite(oracle_mod, sf, cr):
    """suite `short`: SF x CR x {deadbeef x5, 88 x1, ffff x10}; harness ctor cr=4
    (qa_testsuite.py:232) -> exercises the stale-header-CR path for cr<=2."""
    cfg = synth.TxConfig(sf=sf, cr=cr, crc=True, reduced_rate=(sf > 10))
    reps = 1.0 if sf <= 8 else 0.4
    payloads = []
    for hexs, times in SHORT_PAYLOADS:
        payloads += [bytes.fromhex(hexs)] * max(1, int(times * reps))
    st = synth.build_stream(payloads, cfg, rng=np.random.default_rng(100 * sf + cr))
    for mode in (0, 2):
        o = _run(oracle_mod, st, cfg, mode)
        got = [f[15:] for f in o.frames()]
        assert got == [synth.expected_frame_tail(p, cfg) for p in payloads], (sf, cr, mode)


@pytest.mark.slow
@pytest.mark.parametrize("sf", [11, 12])
def test_short_suite_high_sf(oracle_mod, sf):
    cfg = synth.TxConfig(sf=sf, cr=4, crc=True, reduced_rate=True)    # reduced_rate=(sf>10), qa_testsuite.py:228-231
    payloads = [bytes.fromhex("deadbeef"), bytes.fromhex("88")]
    st = synth.build_stream(payloads, cfg, rng=np.random.default_rng(sf))
    o = _run(oracle_mod, st, cfg, 0)
    assert [f[15:] for f in o.frames()] == [synth.expected_frame_tail(p, cfg) for p in payloads]


@pytest.mark.parametrize("sf", [7, 9])
def test_decode_long(oracle_mod, sf):
    """suite `decode_long`: payload 00..fe (255 B), CR4/8 (generate_test_suites.py:157-166)."""
    cfg = synth.TxConfig(sf=sf, cr=4, crc=True)
    payload = bytes(range(255))
    st = synth.build_stream([payload], cfg, rng=np.random.default_rng(sf))
    for mode in (0, 1):
        o = _run(oracle_mod, st, cfg, mode)
        fr = o.frames()
        assert len(fr) == 1 and fr[0][15:] == synth.expected_frame_tail(payload, cfg)


def test_positions_and_state_sequence(oracle_mod):
    """SYNC lands on a symbol boundary; header starts sps + sps + sps/4 after the
    first downchirp (decoder_impl.cc:816,822) -- SURVEY Appendix C."""
    cfg = synth.TxConfig(sf=7, cr=4)
    st = synth.build_stream([b"\x01\x02\x03"], cfg, gaps=[3000])
    o = _run(oracle_mod, st, cfg, 0)
    o2 = oracle_mod.Oracle(sf=7); o2.enable_trace(); o2.run(st.iq)
    tr = o2.trace()
    assert o.frame_positions() == st.header_starts
    sync = [t for t in tr if t[0] == 1]
    assert (sync[-1][1] + sync[-1][2] - st.frame_starts[0]) % cfg.sps == 0
    hdr = [t for t in tr if t[0] == 4]
    assert len(hdr) == 8 and hdr[0][1] == st.header_starts[0]
    assert [t[3] for t in hdr] == [(s - 1) % 128 for s in st.shifts[0][0]]     # bin_idx = s-1 (M2)


def test_gradient_quirk_and_compat_switch(oracle_mod):
    """M2: a symbol with shift 0 (bin N-1) demodulates to 0 in the default path.
    At CR4/5 the wrong bit is not repaired; FFT(-1) decodes, FFT_COMPAT == default."""
    cfg = synth.TxConfig(sf=7, cr=1, crc=False)
    rng = np.random.default_rng(5)
    found = None
    for _ in range(400):
        p = bytes(rng.integers(0, 256, 12, dtype=np.uint8))
        h, q = synth.encode_shifts(p, cfg)
        if 0 in q and 0 not in h:
            found = p
            break
    assert found is not None
    st = synth.build_stream([found], cfg, gaps=[2500])
    want = synth.expected_frame_tail(found, cfg)
    grad = _run(oracle_mod, st, cfg, 0, ctor_cr=1).frames()
    fft = _run(oracle_mod, st, cfg, 1, ctor_cr=1).frames()
    compat = _run(oracle_mod, st, cfg, 2, ctor_cr=1).frames()
    assert len(grad) == len(fft) == len(compat) == 1
    assert fft[0][15:] == want
    assert grad[0][15:] != want
    assert compat[0] == grad[0]


def test_fft_and_gradient_bins_on_ideal_symbols(oracle_mod):
    """Exhaustive over shifts (SURVEY M2 probe): FFT == s; gradient == s-1, s=0 -> 0."""
    for sf in (7, 8):
        cfg = synth.TxConfig(sf=sf)
        o = oracle_mod.Oracle(sf=sf)
        up = synth.base_upchirp(cfg)
        ar = np.arange(cfg.sps)
        for s in range(cfg.nbins):
            sym = up[(ar + s * cfg.decim) % cfg.sps]
            assert o.get_shift_fft(sym) == s
            assert o.max_frequency_gradient_idx(sym) == ((s - 1) % cfg.nbins if s else 0)


def test_scheduler_tail_rule(oracle_mod):
    """work() only runs while 2*sps items remain (set_output_multiple, :91): a
    frame whose last symbol starts inside the final 2*sps is not published."""
    cfg = synth.TxConfig(sf=7, cr=4)
    st = synth.build_stream([b"abcd"], cfg, gaps=[2048], tail_symbols=3.0)
    assert len(_run(oracle_mod, st, cfg).frames()) == 1
    cut = st.iq[: len(st.iq) - 3 * cfg.sps + cfg.sps // 2]      # < 2*sps after last symbol start
    st2 = synth.SynthStream(iq=cut, payloads=st.payloads, frame_starts=st.frame_starts, header_starts=st.header_starts)
    assert len(_run(oracle_mod, st2, cfg).frames()) == 0


def test_chunked_feed_equals_single_feed(oracle_mod):
    cfg = synth.TxConfig(sf=8, cr=3)
    payloads = [b"hello", b"world!!", b"x"]
    st = synth.build_stream(payloads, cfg, rng=np.random.default_rng(3))
    whole = _run(oracle_mod, st, cfg).frames()
    assert [f[15:] for f in whole] == [synth.expected_frame_tail(p, cfg) for p in payloads]


def test_awgn_moderate_snr(oracle_mod):
    """Front-end gates (0.90 / 0.96, :755,:792) hold at high SNR (>= 24 dB)."""
    cfg = synth.TxConfig(sf=7, cr=4)
    rng = np.random.default_rng(11)
    payloads = [bytes(rng.integers(0, 256, 16, dtype=np.uint8)) for _ in range(4)]
    st = synth.build_stream(payloads, cfg, rng=rng, noise_sigma=10 ** (-30 / 20.0))
    for mode in (0, 1):
        got = [f[15:] for f in _run(oracle_mod, st, cfg, mode).frames()]
        assert got == [synth.expected_frame_tail(p, cfg) for p in payloads]


def test_implicit_mode_runs(oracle_mod):
    """Implicit header: 8 reduced-rate symbols carry payload codewords, packet ends
    when the symbol energy halves (decoder_impl.cc:828-829, 861-864)."""
    cfg = synth.TxConfig(sf=8, cr=4, crc=False, implicit=True)
    payload = bytes(range(1, 13))
    st = synth.build_stream([payload], cfg, gaps=[5000])
    o = _run(oracle_mod, st, cfg, 0)
    fr = o.frames()
    assert len(fr) == 1
    assert fr[0][18:18 + len(payload)] == payload


def test_gradient_fragility_after_quirk_symbol(oracle_mod):
    """Documented divergence: the default gradient path decodes packet 19 of this clean
    stream wrong (window 1 sample late after an s=0 symbol), both FFT modes decode it."""
    cfg = synth.TxConfig(sf=7, cr=4)
    rng = np.random.default_rng(4242 + 40)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(1, 48)), dtype=np.uint8)) for _ in range(40)]
    gaps = [int(g) for g in rng.integers(0, 7 * cfg.sps, len(payloads))]
    gaps[5] = 0; gaps[6] = 1; gaps[7] = cfg.sps // 2; gaps[8] = 2 * cfg.sps + 3
    st = synth.build_stream(payloads, cfg, gaps=gaps)
    expect = [synth.expected_frame_tail(p, cfg) for p in payloads]
    bad = {}
    for mode in (0, 1, 2):
        fr = _run(oracle_mod, st, cfg, mode).frames()
        assert len(fr) == 40
        bad[mode] = [i for i, f in enumerate(fr) if f[15:] != expect[i]]
    assert bad == {0: [19], 1: [], 2: []}
