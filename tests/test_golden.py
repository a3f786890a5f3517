"""Committed golden fixtures (tests/golden/, made by make_golden.py): the oracle and
the transmit model must keep reproducing them (CPU); the HIP path must too (GPU)."""
import json
import os

import numpy as np
import pytest

from gr_lora_amd import synth

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLD = json.load(open(os.path.join(HERE, "golden.json")))


def _stream(case):
    cfg = synth.TxConfig(sf=case["sf"], cr=case["cr"], crc=True)
    payloads = [bytes.fromhex(p) for p in case["payloads"]]
    st = synth.build_stream(payloads, cfg, gaps=case["gaps"])
    assert st.iq.size == case["n_items"]
    assert [[h, q] for h, q in st.shifts] == case["shifts"]
    return cfg, st


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
    for mode in (0, 1, 2):
        o = oracle_mod.Oracle(sf=case["sf"], cr=4, crc=True, demod=mode)
        o.enable_trace()
        o.run(st.iq)
        g = case["modes"][str(mode)]
        assert [f.hex() for f in o.frames()] == g["frames"]
        assert o.frame_positions() == g["header_pos"]
        assert [t[3] for t in o.trace() if t[0] in (4, 5)] == g["bins"]


@pytest.mark.gpu
def test_gpu_reproduces_golden():
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
        for mode in (0, 1, 2):
            h = capi.Handle(sf=case["sf"], cr=4, demod=mode, flags=capi.FLAG_TRACE)
            h.decode_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size], 0)
            got = h.drain()
            g = case["modes"][str(mode)]
            assert [f.hex() for f, _ in got] == g["frames"]
            assert [i.header_pos for _, i in got] == g["header_pos"]
            assert [t[3] for t in h.trace() if t[0] in (4, 5)] == g["bins"]
            h.close()
