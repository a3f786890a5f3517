"""Committed golden fixtures (tests/golden/golden.json, written by tests/golden/make_golden.py FROM THE REFERENCE:
/root/reference/lib/decoder_impl.cc compiled unmodified, oracle/_ref).  The restated oracle and the transmit model
must keep reproducing them (CPU, no reference needed at test time); the HIP path must too (GPU).

  case["ref"]   frames, header positions and the complete work() trace of the reference's shipped receive path
                (gradient demodulator) - compared with DEMOD_GRAD;
  case["fft"]   the reference's get_shift_fft at ground-truth symbol offsets - compared with the FFT demodulator;
  case["modes"] FFT / FFT_COMPAT receive paths, which upstream does not have (decoder_impl.cc:500 is commented out):
                oracle-made and labelled so.
"""
import json
import os

import numpy as np
import pytest

from gr_lora_amd import synth

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLD = json.load(open(os.path.join(HERE, "golden.json")))


def _kw(case):
    return dict(sf=case["sf"], cr=case["cr"], crc=case["crc"], implicit=case["implicit"], reduced_rate=case["reduced_rate"],
                disable_drift_correction=case["disable_drift_correction"])


def _stream(case):
    cfg = synth.TxConfig(sf=case["sf"], cr=case["cr"], crc=case["crc"], reduced_rate=case["reduced_rate"], implicit=case["implicit"])
    payloads = [bytes.fromhex(p) for p in case["payloads"]]
    st = synth.build_stream(payloads, cfg, gaps=case["gaps"])
    assert st.iq.size == case["n_items"]
    assert [[h, q] for h, q in st.shifts] == case["shifts"]
    return cfg, st


def test_fixture_provenance():
    assert "oracle/_ref" in GOLD["generated_by"]
    assert all(c["ref"]["source"].startswith("reference") and c["fft"]["source"].startswith("reference") for c in GOLD["cases"])
    assert {(c["sf"], c["cr"]) for c in GOLD["cases"]} >= {(sf, cr) for sf in (7, 8, 9, 10) for cr in (1, 2, 3, 4)} | {(11, 4), (12, 1)}
    assert any(c["implicit"] for c in GOLD["cases"]) and any(c["disable_drift_correction"] for c in GOLD["cases"])


def test_iq_fixture_on_oracle(oracle_mod):
    iq = np.fromfile(os.path.join(HERE, GOLD["iq_file"]["name"]), dtype=np.complex64)
    for mode in (0, 1, 2):
        fr = oracle_mod.decode_stream(iq, demod=mode, sf=7, cr=4)
        assert [f.hex() for f in fr] == GOLD["iq_file"]["frames"]
        assert all(f[15:].hex() == GOLD["readme_known_answer"] for f in fr)


@pytest.mark.parametrize("idx", range(len(GOLD["cases"])))
def test_oracle_reproduces_golden(oracle_mod, idx):
    case = GOLD["cases"][idx]
    cfg, st = _stream(case)
    o = oracle_mod.Oracle(demod=oracle_mod.DEMOD_GRAD, **_kw(case))
    o.enable_trace()
    o.run(st.iq)
    g = case["ref"]
    assert [f.hex() for f in o.frames()] == g["frames"]
    assert o.frame_positions() == g["header_pos"]
    assert [list(t[:5]) for t in o.trace()] == g["trace"]
    assert o.demod_at(st.iq, case["fft"]["offsets"], oracle_mod.DEMOD_FFT).tolist() == case["fft"]["shifts"]
    for mode in (1, 2):
        o = oracle_mod.Oracle(demod=mode, **_kw(case))
        o.enable_trace()
        o.run(st.iq)
        g = case["modes"][str(mode)]
        assert [f.hex() for f in o.frames()] == g["frames"]
        assert o.frame_positions() == g["header_pos"]
        assert [t[3] for t in o.trace() if t[0] in (4, 5)] == g["bins"]


@pytest.mark.gpu
def test_gpu_reproduces_golden():
    """The HIP path against what the compiled reference did: frames, header positions, every work() step."""
    import torch
    assert torch.cuda.is_available()
    from gr_lora_amd import capi
    iq = np.fromfile(os.path.join(HERE, GOLD["iq_file"]["name"]), dtype=np.complex64)
    for mode in (0, 1, 2):
        h = capi.Handle(sf=7, cr=4, demod=mode)
        dev = torch.from_numpy(iq.view(np.float32)).cuda()
        h.decode_device(dev.data_ptr(), iq.size, [0], [iq.size], 0)
        assert [f.hex() for f, _ in h.drain()] == GOLD["iq_file"]["frames"]
        h.close()
    for case in GOLD["cases"]:
        cfg, st = _stream(case)
        dev = torch.from_numpy(st.iq.view(np.float32)).cuda()
        h = capi.Handle(demod=capi.DEMOD_GRAD, flags=capi.FLAG_TRACE, **_kw(case))
        h.decode_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size], 0)
        got = h.drain()
        g = case["ref"]
        tag = (case["sf"], case["cr"], case["implicit"], case["disable_drift_correction"])
        assert [f.hex() for f, _ in got] == g["frames"], tag
        assert [i.header_pos for _, i in got] == g["header_pos"], tag      # (SF11 / SF12 too: strict SYNC, tests/test_gpu_strict_sync.py)
        assert [list(t[:5]) for t in h.trace()] == g["trace"], tag
        shifts = h.demod_symbols_device(dev.data_ptr(), st.iq.size, case["fft"]["offsets"], capi.DEMOD_FFT)
        assert shifts.tolist() == case["fft"]["shifts"], tag
        h.close()
        for mode in (1, 2):
            h = capi.Handle(demod=mode, flags=capi.FLAG_TRACE, **_kw(case))
            h.decode_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size], 0)
            got = h.drain()
            g = case["modes"][str(mode)]
            assert [f.hex() for f, _ in got] == g["frames"], tag
            assert [i.header_pos for _, i in got] == g["header_pos"], tag
            assert [t[3] for t in h.trace() if t[0] in (4, 5)] == g["bins"], tag
            h.close()
