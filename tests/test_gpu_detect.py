"""FFT-domain preamble detection on the device (SURVEY 8(f) N4 remainder): lora_hip_window_stats_device and
lora_hip_detect_preambles_device against their float64 definition (oracle/preamble_oracle.py), detection probability against
ground truth at each spreading factor's sensitivity - tens of dB under the reference's own gates (decoder_impl.cc:755, :792) -
and BASELINE config 5 END TO END: no ground-truth timing, the detector's header position feeds lora_hip_demod_symbols_device
and the symbol bins are held to the transmitted ones."""
import numpy as np
import pytest

from gr_lora_amd import synth
from oracle import preamble_oracle as PO

pytestmark = pytest.mark.gpu


def _dev(iq):
    import torch
    return torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).cuda()


def _stream(sf, snr_db, cfo_hz, seed, n=3, length=10):
    cfg = synth.TxConfig(sf=sf, cr=4, reduced_rate=(sf > 10))
    rng = np.random.default_rng(seed)
    payloads = [bytes(rng.integers(0, 256, length, dtype=np.uint8)) for _ in range(n)]
    gaps = [int(g) for g in rng.integers(2 * cfg.sps, 7 * cfg.sps, n)]
    st = synth.build_stream(payloads, cfg, gaps=gaps, rng=rng, tail_symbols=3.0, noise_sigma=(synth.awgn_sigma_for_snr(snr_db, cfg) if snr_db is not None else 0.0), cfo_hz=cfo_hz)
    return cfg, st


@pytest.mark.parametrize("sf,snr_db", [(7, 0.0), (9, -8.0), (11, -12.0), (12, -15.0)])
def test_window_stats_and_detection_equal_the_definition(oracle_mod, sf, snr_db):
    from gr_lora_amd import capi
    cfg, st = _stream(sf, snr_db, 1500.0, seed=70 + sf)
    o = oracle_mod.Oracle(sf=sf, reduced_rate=(sf > 10))
    down = o.table(0).view(np.complex64)
    dev = _dev(st.iq)
    h = capi.Handle(sf=sf, reduced_rate=(sf > 10))
    offs = np.arange(0, st.iq.size - cfg.sps, cfg.sps)[:48]
    got = h.window_stats_device(dev.data_ptr(), st.iq.size, offs)
    strong = 0
    for p, g in zip(offs, got):
        w = PO.window_stats(st.iq, int(p), down, cfg.nbins)
        for gi, wi in ((g[0:3], w[0:3]), (g[3:6], w[3:6])):
            assert abs(gi[2] - wi[2]) <= 2e-3 * wi[2]                                   # total power
            if wi[1] * (cfg.nbins - 1) / (wi[2] - wi[1]) >= 2 * PO.default_threshold(cfg.nbins):   # a real peak: same bin, same power
                assert gi[0] == wi[0] and abs(gi[1] - wi[1]) <= 2e-3 * wi[1], (p, gi, wi)
                strong += 1
            else:                                                                         # noise: the largest bin's power agrees
                assert abs(gi[1] - wi[1]) <= 2e-3 * wi[1], (p, gi, wi)
    assert strong >= 12
    want = PO.detect(st.iq, down, cfg.nbins)
    have = h.detect_preambles_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size])
    h.close()
    assert len(want) == 3
    assert [(d["header_pos"], d["bin"], d["sfd_index"], d["run_len"], d["delta"]) for d in have] == [(d["header_pos"], d["bin"], d["sfd_index"], d["run_len"], d["delta"]) for d in want]
    for a, b in zip(have, want):
        assert a["cfo_bins"] == b["cfo_bins"] and abs(a["pmr"] - b["pmr"]) <= 5e-3 * b["pmr"] and abs(a["cfo_hz"] - b["cfo_bins"] * cfg.bw / cfg.nbins) < 1e-2


# in-band SNR LoRa specifies as each spreading factor's demodulation limit (SX127x data sheet)
SENSITIVITY_DB = {7: -7.5, 8: -10.0, 9: -12.5, 10: -15.0, 11: -17.5, 12: -20.0}


@pytest.mark.parametrize("sf,n", [(7, 40), (8, 40), (9, 30), (10, 20), (11, 10), (12, 6)])
def test_detection_probability_vs_snr(sf, n):
    """Pd against ground truth over SNR.  Single-symbol spectra (no accumulation over the preamble): every packet is acquired
    from 5 dB above the SF's demodulation limit - SF7 at -2.5 dB ... SF12 at -15 dB, where the reference's own gates
    (decoder_impl.cc:755, :792) acquire nothing (tests/test_preamble_oracle.py; SURVEY M7: 0 of 6 at <= 20 dB) - most of them
    2.5 dB above it, and the curve does not fall with rising SNR."""
    from gr_lora_amd import capi
    h = capi.Handle(sf=sf, reduced_rate=(sf > 10))
    pd = []
    for up_db in (2.5, 5.0, 15.0):
        cfg, st = _stream(sf, SENSITIVITY_DB[sf] + up_db, -1100.0, seed=300 + sf + int(2 * up_db), n=n, length=6)
        dev = _dev(st.iq)
        det = h.detect_preambles_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size])
        cfo_bins = -1100.0 / (cfg.bw / cfg.nbins)
        hits = sum(any(abs(d["header_pos"] + cfo_bins * cfg.decim - t) <= 1.5 * cfg.decim for d in det) for t in st.header_starts)
        assert len(det) - hits <= max(1, n // 10), (sf, up_db, len(det), hits)     # false alarms
        pd.append(hits / n)
    h.close()
    assert pd[0] >= 0.7 and pd[1] >= 0.95 and pd[2] == 1.0, (sf, pd)


def test_no_false_alarm_on_noise_many_streams():
    from gr_lora_amd import capi
    cfg = synth.TxConfig(sf=9)
    rng = np.random.default_rng(12)
    n = 8 * 60 * cfg.sps
    noise = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    dev = _dev(noise)
    h = capi.Handle(sf=9)
    assert h.detect_preambles_device(dev.data_ptr(), n, [k * 60 * cfg.sps for k in range(8)], [60 * cfg.sps] * 8) == []
    h.close()


@pytest.mark.parametrize("reduced_rate", [False, True])
def test_config5_end_to_end_without_ground_truth(reduced_rate):
    """BASELINE config 5: SF12, 255-byte payload 00..fe, carrier offset, AWGN at -10 dB in-band - where the reference acquires
    nothing (tests/test_preamble_oracle.py).  Detector -> header position -> get_shift_fft per symbol: bins against the
    TRANSMITTED symbols within +-1."""
    from gr_lora_amd import capi
    cfg = synth.TxConfig(sf=12, cr=4, reduced_rate=reduced_rate)
    rng = np.random.default_rng(5)
    cfo = float(rng.uniform(-cfg.bw / 4, cfg.bw / 4))
    st = synth.build_stream([bytes(range(255))], cfg, gaps=[3 * cfg.sps + 4321], rng=rng, noise_sigma=synth.awgn_sigma_for_snr(-10.0, cfg), cfo_hz=cfo, tail_symbols=2.5)
    n_sym = min(8 + len(st.shifts[0][1]), 160)
    iq = st.iq[: st.header_starts[0] + (n_sym + 2) * cfg.sps]
    dev = _dev(iq)
    h = capi.Handle(sf=12, cr=4, reduced_rate=reduced_rate)
    det = h.detect_preambles_device(dev.data_ptr(), iq.size, [0], [iq.size])
    assert len(det) == 1
    d = det[0]
    assert abs(d["cfo_hz"] - cfo) <= 0.75 * cfg.bw / cfg.nbins
    assert abs(d["header_pos"] + d["cfo_bins"] * cfg.decim - st.header_starts[0]) <= 1.5 * cfg.decim
    offs = d["header_pos"] + np.arange(n_sym) * cfg.sps
    got = h.demod_symbols_device(dev.data_ptr(), iq.size, offs, 1).astype(np.int64)
    h.close()
    truth = np.array((st.shifts[0][0] + st.shifts[0][1])[:n_sym])     # on the aligned clock the carrier offset is gone: the bins ARE the shifts
    e = np.abs(got - truth)
    e = np.minimum(e, cfg.nbins - e)
    assert (e == 0).mean() >= 0.97 and (e <= 1).all(), (np.nonzero(e > 0)[0][:10], e.max())   # (with the sub-bin refinement: the bins themselves)


@pytest.mark.parametrize("sf,demod", [(7, 2), (7, 0), (8, 1), (9, 2), (10, 0), (12, 2)])
def test_decode_at_given_headers_equals_normal_decode(oracle_mod, sf, demod):
    """lora_hip_decode_at_headers_device on clean packets at their true header positions (the positions the normal receive path
    reports): same frames (behind the loratap header: its SNR byte needs DETECT), every kernel family."""
    from gr_lora_amd import capi
    cfg = synth.TxConfig(sf=sf, cr=4, reduced_rate=(sf > 10))
    rng = np.random.default_rng(40 + sf)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(3, 40)), dtype=np.uint8)) for _ in range(4)]
    st = synth.build_stream(payloads, cfg, rng=rng, gap_symbols=(2.0, 6.0))
    dev = _dev(st.iq)
    kw = dict(sf=sf, cr=4, reduced_rate=(sf > 10), demod=demod)
    h = capi.Handle(**kw)
    h.decode_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size], 0)
    normal = [(b[15:], i.header_pos) for b, i in h.drain()]
    assert len(normal) == 4
    h.decode_at_headers_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size], [(0, p) for _b, p in normal])
    got = [(b[15:], i.header_pos) for b, i in h.drain()]
    h.close()
    assert got == normal


@pytest.mark.parametrize("reduced_rate", [False, True])
def test_config5_frames_end_to_end(reduced_rate):
    """BASELINE config 5 all the way to bytes: detector -> lora_hip_decode_at_headers_device (FFT demodulator, drift correction off:
    fine_sync's ifreq correlation is a time-domain estimator like the reference's gates) -> the 255-byte payload 00..fe, at -10 dB
    in-band with a carrier offset, where the reference's receive path publishes nothing."""
    from gr_lora_amd import capi
    cfg = synth.TxConfig(sf=12, cr=4, reduced_rate=reduced_rate)
    rng = np.random.default_rng(5)
    cfo = float(rng.uniform(-cfg.bw / 4, cfg.bw / 4))
    st = synth.build_stream([bytes(range(255))], cfg, gaps=[3 * cfg.sps + 4321], rng=rng, noise_sigma=synth.awgn_sigma_for_snr(-10.0, cfg), cfo_hz=cfo, tail_symbols=2.5)
    dev = _dev(st.iq)
    h = capi.Handle(sf=12, cr=4, reduced_rate=reduced_rate, demod=capi.DEMOD_FFT, disable_drift_correction=True)
    det = h.detect_preambles_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size])
    assert len(det) == 1
    h.decode_at_headers_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size], det)
    frames = h.drain()
    h.close()
    assert len(frames) == 1
    body = frames[0][0][15:]
    want = synth.expected_frame_tail(bytes(range(255)), cfg)
    wrong = sum(a != b for a, b in zip(body, want)) + abs(len(body) - len(want))
    assert wrong == 0, (wrong, len(body), len(want))
