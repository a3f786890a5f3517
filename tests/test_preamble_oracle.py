"""oracle/preamble_oracle.py (the float64 definition of the FFT-domain preamble detector) on its own: what it
acquires, how exactly, and that the reference's time-domain gates (decoder_impl.cc:340-366 / :755) cannot at those levels."""
import numpy as np
import pytest

from gr_lora_amd import synth
from oracle import preamble_oracle as PO


def _stream(sf, snr_db, cfo_hz, seed, n=3):
    cfg = synth.TxConfig(sf=sf, cr=4, reduced_rate=(sf > 10))
    rng = np.random.default_rng(seed)
    payloads = [bytes(rng.integers(0, 256, 10, dtype=np.uint8)) for _ in range(n)]
    gaps = [int(g) for g in rng.integers(2 * cfg.sps, 7 * cfg.sps, n)]
    st = synth.build_stream(payloads, cfg, gaps=gaps, rng=rng, tail_symbols=3.0, noise_sigma=(synth.awgn_sigma_for_snr(snr_db, cfg) if snr_db is not None else 0.0), cfo_hz=cfo_hz)
    return cfg, st


@pytest.mark.parametrize("sf,snr_db", [(7, -5.0), (8, -7.0), (9, -10.0), (10, -12.0)])
def test_acquires_below_zero_db_where_the_reference_gate_cannot(oracle_mod, sf, snr_db):
    cfo = 2000.0 if sf < 10 else -900.0
    cfg, st = _stream(sf, snr_db, cfo, seed=40 + sf)
    o = oracle_mod.Oracle(sf=sf, reduced_rate=(sf > 10))
    down = o.table(0).view(np.complex64)
    det = PO.detect(st.iq, down, cfg.nbins)
    assert len(det) == len(st.header_starts)
    cfo_bins = cfo / (cfg.bw / cfg.nbins)
    for d, truth in zip(det, st.header_starts):
        # the aligned clock absorbs the carrier offset: header_pos = truth - cfo_bins * D, to the resolution of one bin (D samples)
        assert abs(d["header_pos"] + cfo_bins * cfg.decim - truth) <= cfg.decim + 1, (d, truth)
        assert abs(d["cfo_bins"] - cfo_bins) <= 0.75
    # the reference's preamble gate (autocorrelation of adjacent symbols >= 0.90, :755) on the same preambles: never reached
    best = max(o.detect_preamble_autocorr(st.iq[p:p + 2 * cfg.sps]) for h in st.header_starts for p in range(h - 10 * cfg.sps, h - 5 * cfg.sps, cfg.sps // 4))
    assert best < 0.90
    # and its whole receive path publishes nothing
    o2 = oracle_mod.Oracle(sf=sf, reduced_rate=(sf > 10), demod=2)
    o2.run(st.iq)
    assert o2.frames() == []


def test_no_false_alarm_on_noise_and_clean_timing():
    cfg = synth.TxConfig(sf=8)
    rng = np.random.default_rng(3)
    noise = (rng.standard_normal(200 * cfg.sps) + 1j * rng.standard_normal(200 * cfg.sps)).astype(np.complex64)
    down = np.conj(synth.base_upchirp(cfg)).astype(np.complex64)
    assert PO.detect(noise, down, cfg.nbins) == []
    cfg, st = _stream(8, None, 0.0, seed=9, n=4)
    det = PO.detect(st.iq, down, cfg.nbins)
    assert len(det) == 4
    for d, truth in zip(det, st.header_starts):
        assert -cfg.decim <= d["header_pos"] - truth <= 1 and d["cfo_bins"] == 0.0
