"""Pins the restated CPU oracle (oracle/lora_oracle.c) to the REFERENCE ITSELF: /root/reference/lib/decoder_impl.cc
compiled unmodified against stand-in headers (oracle/ref_build/ -> oracle/_ref/libref_decoder.so) and driven by a
scheduler loop.  For every cell of the reference's own suite matrices (`short`: SF7-12 x CR4/5-4/8, `decode_long`:
255-byte payloads; apps/generate_test_suites.py:157-203), with drift correction on and off, explicit and implicit
header, clean and noisy, the two must publish IDENTICAL frames (all bytes, loratap included), at identical header
positions, through an identical sequence of work() calls (state, input position, consume_each, bin, d_fine_sync) and
bit-identical decision values (autocorrelation, sliding maximum, SFD correlation).

What this does NOT pin (stated in oracle/lora_oracle.h): the stand-ins for VOLK (generic sequential loops) and liquid
(double DFT; Hamming(8,4) nearest codeword, lowest symbol on ties) are this repository's reading of those libraries.
"""
import math

import numpy as np
import pytest

from gr_lora_amd import synth
from oracle import ref as R

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built and /root/reference absent")

HAMMING84 = [0x00, 0xd2, 0x55, 0x87, 0x99, 0x4b, 0xcc, 0x1e, 0xe1, 0x33, 0xb4, 0x66, 0x78, 0xaa, 0x2d, 0xff]


def _stream(sf, cr, n_packets, seed, implicit=False, crc=True, noise_db=None, lengths=(1, 24), payloads=None, cfo_hz=0.0):
    rng = np.random.default_rng(seed)
    cfg = synth.TxConfig(sf=sf, cr=cr, crc=crc, reduced_rate=(sf > 10), implicit=implicit)
    if payloads is None:
        payloads = [bytes(rng.integers(0, 256, int(rng.integers(*lengths)), dtype=np.uint8)) for _ in range(n_packets)]
    sigma = synth.awgn_sigma_for_snr(noise_db, cfg) if noise_db is not None else 0.0
    return cfg, synth.build_stream(payloads, cfg, rng=rng, noise_sigma=sigma, cfo_hz=cfo_hz)


def _same_value(a, b):
    return (math.isnan(a) and math.isnan(b)) or a == b


def _assert_identical(oracle_mod, iq, **kw):
    r = R.Reference(**kw)
    r.enable_trace()
    n_r = r.run(iq)
    o = oracle_mod.Oracle(demod=oracle_mod.DEMOD_GRAD, **kw)   # the reference's shipped demodulator (:499)
    o.enable_trace()
    n_o = o.run(iq)
    assert n_r == n_o
    fr, fo = r.frames(), o.frames()
    assert [f.hex() for f in fr] == [f.hex() for f in fo]
    assert r.frame_positions() == o.frame_positions()
    tr, to = r.trace(), o.trace()
    assert len(tr) == len(to)
    for i, (a, b) in enumerate(zip(tr, to)):
        assert a[:5] == b[:5], (i, a, b)
        assert _same_value(a[5], b[5]), (i, a, b)
    return r, fr


@pytest.mark.parametrize("sf", [7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("cr", [1, 2, 3, 4])
def test_short_matrix_identical(oracle_mod, sf, cr):
    """suite `short` geometry: every SF x CR, explicit header, CRC on, reduced rate for SF > 10 (qa_testsuite.py:228-231)."""
    n = 3 if sf <= 9 else (2 if sf == 10 else 1)
    cfg, st = _stream(sf, cr, n, seed=100 * sf + cr, lengths=(1, 24) if sf <= 10 else (1, 8))
    r, frames = _assert_identical(oracle_mod, st.iq, sf=sf, cr=cr, crc=True, reduced_rate=(sf > 10))
    assert len(frames) == n
    # the banner is the reference's (:93-96)
    assert "Bins per symbol: \t%d" % (1 << sf) in r.stdout() and "Samples per symbol: \t%d" % (8 << sf) in r.stdout()


@pytest.mark.parametrize("sf,cr", [(7, 4), (7, 1), (8, 2), (8, 3), (9, 4), (10, 1)])
def test_drift_correction_disabled_identical(oracle_mod, sf, cr):
    """decoder::make(..., disable_drift_correction=true): d_enable_fine_sync = false (:90,:501)."""
    cfg, st = _stream(sf, cr, 3, seed=7 * sf + cr)
    r, frames = _assert_identical(oracle_mod, st.iq, sf=sf, cr=cr, disable_drift_correction=True)
    assert len(frames) == 3
    assert "Warning: clock drift correction disabled" in r.stdout()


@pytest.mark.parametrize("sf,cr,crc", [(7, 4, True), (7, 1, False), (8, 3, True), (9, 2, False), (10, 4, True), (11, 3, True)])
def test_implicit_header_identical(oracle_mod, sf, cr, crc):
    """implicit header: no header parse, payload ends when the symbol energy halves (:828-829,:861-864)."""
    cfg, st = _stream(sf, cr, 2 if sf < 11 else 1, seed=11 * sf + cr, implicit=True, crc=crc, lengths=(4, 20) if sf < 11 else (2, 6))
    r, frames = _assert_identical(oracle_mod, st.iq, sf=sf, cr=cr, crc=crc, implicit=True, reduced_rate=(sf > 10))
    assert len(frames) >= 1
    assert "CR: \t\t%d" % cr in r.stdout()


@pytest.mark.parametrize("sf,cr,snr", [(7, 4, 40), (7, 2, 35), (7, 3, 33), (8, 3, 36), (8, 1, 34), (9, 1, 38), (9, 4, 34), (10, 2, 36)])
def test_noisy_streams_identical(oracle_mod, sf, cr, snr):
    """AWGN near the reference's own acquisition limit: failed correlations, lost sync, non-zero fine sync."""
    cfg, st = _stream(sf, cr, 6 if sf < 9 else 3, seed=sf * 7 + cr, noise_db=snr)
    _assert_identical(oracle_mod, st.iq, sf=sf, cr=cr)


@pytest.mark.parametrize("sf,cr,cfo", [(7, 4, 800.0), (8, 2, -1500.0), (9, 3, 400.0)])
def test_frequency_offset_identical(oracle_mod, sf, cr, cfo):
    cfg, st = _stream(sf, cr, 3, seed=sf + cr, cfo_hz=cfo, noise_db=40)
    _assert_identical(oracle_mod, st.iq, sf=sf, cr=cr)


@pytest.mark.parametrize("sf", [7, 8, 9, 10, pytest.param(11, marks=pytest.mark.slow), pytest.param(12, marks=pytest.mark.slow)])
def test_decode_long_identical(oracle_mod, sf):
    """suite `decode_long`: 255-byte payload 00..fe, CR4/8 (apps/generate_test_suites.py:157-170)."""
    cfg, st = _stream(sf, 4, 1, seed=sf, payloads=[bytes(range(255))])
    r, frames = _assert_identical(oracle_mod, st.iq, sf=sf, cr=4)
    assert len(frames) == 1 and frames[0][15] == 255


def test_stale_header_cr_and_back_to_back_identical(oracle_mod):
    """Header FEC follows the PREVIOUS packet's CR (:655,:833-835): mixed-CR stream, constructor CR differing."""
    rng = np.random.default_rng(3)
    pieces = []
    for cr in (1, 4, 2, 3, 1, 1, 4):
        cfg = synth.TxConfig(sf=7, cr=cr)
        st = synth.build_stream([bytes(rng.integers(0, 256, 9, dtype=np.uint8))], cfg, rng=rng, tail_symbols=0.0)
        pieces.append(st.iq)
    iq = np.concatenate(pieces + [np.zeros(4096, np.complex64)])
    for ctor_cr in (4, 1):
        _assert_identical(oracle_mod, iq, sf=7, cr=ctor_cr)


def test_readme_known_answer_on_the_reference():
    """README.md:75-85 through the compiled reference: bytes and its own console line."""
    cfg = synth.TxConfig(sf=7, cr=4, crc=True, sync_shifts=(24, 32))
    st = synth.build_stream([bytes.fromhex("deadbeef")] * 5, cfg, rng=np.random.default_rng(868))
    r = R.Reference(sf=7, cr=4)
    r.run(st.iq)
    assert [f[15:].hex() for f in r.frames()] == ["049040deadbeef700d"] * 5
    assert r.stdout().count(" 04 90 40 de ad be ef 70 0d") == 5


def test_block_contract():
    L = R.lib()
    assert L.ref_make_smoke() == 1                       # decoder::make links and yields the block
    r = R.Reference(sf=9, cr=2)
    assert L.ref_output_multiple(r.h) == 2 * r.sps       # :91
    assert [L.ref_port_name(r.h, i) for i in range(L.ref_num_ports(r.h))] == [b"frames", b"control"]   # :120-121
    assert [L.ref_in_sig(r.h, k) for k in range(3)] == [1, -1, 8]                                       # :51
    assert (L.ref_sps(r.h), L.ref_bins(r.h), L.ref_bins_hdr(r.h), L.ref_decim(r.h), L.ref_delay_after_sync(r.h)) == \
        (4096, 512, 128, 8, 1024)
    assert L.ref_dt(r.h) == float(np.float32(1.0) / np.float32(1e6))                                    # :77, float division
    assert L.ref_sizeof(0) == 15 and L.ref_sizeof(1) == 3
    with pytest.raises(ValueError):
        R.Reference(sf=5)
    with pytest.raises(ValueError):
        R.Reference(sf=14)


@pytest.mark.parametrize("sf", [7, 9, 12])
def test_tables_bit_identical(oracle_mod, sf):
    """build_ideal_chirps (:141-175): all five tables, every float."""
    r = R.Reference(sf=sf)
    o = oracle_mod.Oracle(sf=sf)
    for which in range(5):
        a, b = r.table(which), o.table(which)
        n = min(a.size, b.size)                          # the oracle pads upchirp_ifreq_v with a guard
        assert n >= (2 if which < 2 else 1) * r.sps
        assert np.array_equal(a[:n].view(np.uint32), b[:n].view(np.uint32)), which
        if which == 4:
            assert a.size == 3 * r.sps


@pytest.mark.parametrize("sf", [7, 8, 10])
def test_float_primitives_identical(oracle_mod, sf):
    """instantaneous_frequency, autocorrelation (+ its side effects), SFD correlation, sliding correlation, fine_sync,
    gradient demodulator: bit-identical on clean, noisy and misaligned windows."""
    rng = np.random.default_rng(sf)
    cfg, st = _stream(sf, 4, 2, seed=sf, noise_db=38)
    r = R.Reference(sf=sf)
    o = oracle_mod.Oracle(sf=sf)
    sps = r.sps
    for _ in range(24):
        p = int(rng.integers(0, st.iq.size - 2 * sps))
        w = st.iq[p:p + 2 * sps]
        a, b = r.instantaneous_frequency(w[:sps]), oracle_mod.instantaneous_frequency(w[:sps])
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        va, vb = r.detect_preamble_autocorr(w), o.detect_preamble_autocorr(w)
        assert _same_value(va, vb)
        va, vb = r.detect_downchirp(w), o.detect_downchirp(w)
        assert _same_value(va, vb)
        assert r.detect_upchirp(w) == o.detect_upchirp(w)
        g = r.max_frequency_gradient_idx(w)
        assert g == o.max_frequency_gradient_idx(w)
        # bin N-1 is excluded: it reads d_upchirp_ifreq_v[3 sps] (one past the vector, :301,:310) - unreachable with the
        # gradient demodulator, whose largest result is N-2 (:479-490)
        for bin_idx, search in ((g, max(8 // 4, 2)), (-1, 32), (r.nbins - 2, 2), (0, 2)):
            assert r.fine_sync(w, bin_idx, search) == o.fine_sync(w, bin_idx, search)


@pytest.mark.parametrize("sf", [7, 8, 9, 10, 11, 12])
def test_get_shift_fft_against_the_reference(oracle_mod, sf):
    """get_shift_fft (:430-464, the north-star demodulator, dead code upstream): the reference's function - with a
    DOUBLE-precision DFT standing in for liquid's - against the oracle's float radix-2 restatement: every shift at
    SF7, a sample of shifts above, clean and at -5 dB in-band with ground-truth timing (the two FFTs may then differ
    by rounding on a near-tie: at most one bin, at most 1 % of the windows)."""
    rng = np.random.default_rng(sf)
    cfg = synth.TxConfig(sf=sf, cr=4)
    up = synth.base_upchirp(cfg)
    r = R.Reference(sf=sf)
    o = oracle_mod.Oracle(sf=sf)
    n, d = cfg.nbins, cfg.decim
    shifts = range(n) if sf == 7 else [0, 1, n // 2 - 1, n // 2, n // 2 + 1, n - 1] + [int(s) for s in rng.integers(0, n, 24 if sf < 11 else 6)]
    for s in shifts:
        sym = np.roll(up, -s * d)
        assert r.get_shift_fft(sym) == s == o.get_shift_fft(sym)
    sigma = synth.awgn_sigma_for_snr(-5.0, cfg)
    off = 0
    trials = 64 if sf < 11 else 12
    for _ in range(trials):
        s = int(rng.integers(0, n))
        noise = (rng.standard_normal(cfg.sps) + 1j * rng.standard_normal(cfg.sps)) * sigma / np.sqrt(2)
        sym = (np.roll(up, -s * d) + noise).astype(np.complex64)
        a, b = r.get_shift_fft(sym), o.get_shift_fft(sym)
        if a != b:
            assert min((a - b) % n, (b - a) % n) <= 1
            off += 1
    assert off <= max(1, trials // 100)


def test_integer_chain_identical(oracle_mod):
    """rotl, deinterleave (:535-565), deshuffle / dewhiten / hamming_decode / extract_data_only (:567-706),
    utilities.h helpers and lib/tables.h - against the reference's own code."""
    L, OL = R.lib(), oracle_mod.lib()
    rng = np.random.default_rng(1)
    for _ in range(500):
        bits, count, size = int(rng.integers(0, 1 << 12)), int(rng.integers(0, 12)), int(rng.integers(5, 13))
        bits &= (1 << size) - 1
        assert L.ref_rotl(bits, count, size) == OL.lora_oracle_rotl(bits, count, size)
    assert [L.ref_hamming_encode_soft(v) for v in range(16)] == HAMMING84 == [OL.lora_oracle_hamming_encode(v) for v in range(16)]
    assert [OL.lora_oracle_hamming84_decode(c) for c in range(256)] == [L.ref_hamming84_decode_stub(c) for c in range(256)]
    # lib/tables.h verbatim == the product's and the oracle's whitening tables
    w = synth._WHITEN
    assert R.prng(0) == bytes(w["LORA_WHITEN_HEADER"]) and len(R.prng(0)) == 13
    assert R.prng(1) == bytes(w["LORA_WHITEN_CR56"]) and len(R.prng(1)) == 516
    assert R.prng(2) == bytes(w["LORA_WHITEN_CR78"]) and len(R.prng(2)) == 518
    r = R.Reference(sf=9)
    for ppm in (5, 6, 7, 8, 9, 10, 11, 12):
        for n_words in (5, 6, 7, 8):
            words = rng.integers(0, 1 << ppm, n_words, dtype=np.uint32)
            out = (np.zeros(ppm, np.uint8))
            OL.lora_oracle_deinterleave(words.ctypes.data, n_words, ppm, out.ctypes.data)
            assert r.deinterleave(words, ppm) == out.tobytes()
    for v in range(256):
        cw, left = r.decode(bytes([v]) * 5, True, 1)       # header path: deshuffle of 5 codewords, prng_header == 0, data bits
        d = OL.lora_oracle_deshuffle_byte(v)
        nib = ((d >> 1) & 1) | (((d >> 2) & 1) << 1) | (((d >> 3) & 1) << 2) | (((d >> 5) & 1) << 3)
        assert cw == bytes([nib << 4 | nib] * 2 + [nib << 4]) and left == 0     # 5 codewords + the appended 0 (:632)


def test_sync_shift_depends_on_volk_summation_order():
    """How far does the reference pin its own timing decisions?  oracle/_ref/libref_decoder_simd.so is the SAME unmodified
    lib/decoder_impl.cc with the VOLK stand-in summing in 8 lanes (what an AVX protokernel does) instead of sequentially.
    Mathematically identical; the published payload bytes are identical too - but the SYNC step's sliding correlation
    (:399-413) ties between adjacent shifts (1e-6 .. 2e-8 relative for SF8 .. SF12, below the resolution of a float sum of
    sps terms), so the winning shift, and with it every later position, moves by one sample with the summation order.
    This is the latitude the device's trace tests grant at those operating points (tests/parity_util.py)."""
    moved = {}
    for sf, snr, seeds in ((8, None, 4), (10, 40.0, 6), (12, None, 3), (12, 40.0, 2)):
        n_sync = n_moved = 0
        for seed in range(seeds):
            rng = np.random.default_rng(seed + 10 * sf)
            cfg = synth.TxConfig(sf=sf, cr=4, reduced_rate=(sf > 10))
            payloads = [bytes(rng.integers(0, 256, int(rng.integers(2, 12)), dtype=np.uint8)) for _ in range(2)]
            sigma = synth.awgn_sigma_for_snr(snr, cfg) if snr else 0.0
            st = synth.build_stream(payloads, cfg, rng=rng, noise_sigma=sigma)
            runs = []
            for simd in (False, True):
                r = R.Reference(sf=sf, cr=4, reduced_rate=(sf > 10), simd=simd)
                r.enable_trace()
                r.run(st.iq)
                runs.append((r.frames(), r.frame_positions(), r.trace()))
            (f0, p0, t0), (f1, p1, t1) = runs
            assert [f[15:] for f in f0] == [f[15:] for f in f1]            # identical PHY header + payload bytes
            assert len(t0) == len(t1) and [a[0] for a in t0] == [b[0] for b in t1]
            assert all(abs(a[1] - b[1]) <= 1 for a, b in zip(t0, t1))       # never more than one sample apart
            assert all(abs(a - b) <= 1 for a, b in zip(p0, p1))
            for a, b in zip(t0, t1):
                if a[0] == 1:                                               # SYNC
                    n_sync += 1
                    n_moved += a[2] != b[2]
        moved[(sf, snr)] = (n_moved, n_sync)
    assert moved[(8, None)][0] == 0            # resolvable at SF8 without noise ...
    assert moved[(12, None)][0] >= 1           # ... rounding decides at SF12 even on a clean signal


def test_cfo_estimate_vs_reference():
    """experimental_determine_cfo (:730-738; unused upstream): the oracle's restatement, mode 0, against the compiled member function."""
    rng = np.random.default_rng(77)
    for sf in (7, 8, 10, 12):
        from oracle import oracle
        o, r = oracle.Oracle(sf=sf), R.Reference(sf=sf)
        for k in range(6):
            x = (rng.standard_normal(o.sps) + 1j * rng.standard_normal(o.sps)).astype(np.complex64)
            if k % 2:
                cfg = synth.TxConfig(sf=sf)
                x = (synth.base_upchirp(cfg) * np.exp(2j * np.pi * (k * 700.0) * np.arange(o.sps) / 1e6) + 0.1 * x).astype(np.complex64)
            a, b = o.determine_cfo(x, 0), r.experimental_determine_cfo(x)
            assert a == b, (sf, k, a, b)
