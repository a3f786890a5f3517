"""the test-suite harness (apps/generate_test_suites.py, apps/qa_testsuite.py).  CPU: report format and
suite synthesis; GPU: a reduced `short_rn` / `decode_long` matrix through file -> filters -> decoder -> UDP -> scorer."""
import importlib.util
import os

import numpy as np
import pytest

from gr_lora_amd import sigmf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "apps", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_summary_report_format():
    """The report text mirrors python/qa_testsuite.py:39-148 (one table per configuration, totals at the end)."""
    qa = _load("qa_testsuite")
    s = qa.TestSummary("short_rn")
    lc = sigmf.LoRaConfig(868.1e6, 7, "4/8", 125000, 8, True, False)
    s.add(qa.TestResult(["deadbeef"] * 4 + ["deadbeee"], lc, qa.Test("deadbeef", 5)))
    s.add(qa.TestResult(["88"], lc, qa.Test("88", 1)))
    s.add(qa.TestResult([], sigmf.LoRaConfig(868.1e6, 8, "4/5", 125000, 8, True, False), qa.Test("ffff", 10)))
    md = s.export_summary("/nonexistent", print_output=False, write_output=False)
    assert "# Test suite: 'short_rn'" in md
    assert md.count("Transmitted payload | :heavy_check_mark: | :hash: | :heavy_division_sign:") == 2
    assert "`deadbeef                      ` |   4 |   5 | 80.00%" in md
    assert "`88                            ` |   1 |   1 | 100.00%" in md
    assert "`ffff                          ` |   0 |  10 | 0.00%" in md
    assert "Total payloads passed: 5 out of 16 (31.25%)" in md
    assert qa.trunc("00" * 40) == "00000000000000.." + "0" * 14 and len(qa.trunc("00" * 40)) == 30


def test_generate_suites_matrix(tmp_path):
    gen = _load("generate_test_suites")
    assert len(gen.SHORT[1]) == 24 and len(gen.DECODE_LONG[1]) == 6 and len(bytes.fromhex(gen.DECODE_LONG[2][0][0])) == 255
    files = gen.generate(str(tmp_path), gen.SHORT, sfs=[7])
    assert len(files) == 12  # 4 coding rates x 3 payload tests
    meta = sigmf.read_meta(files[0] + ".sigmf-meta")
    assert meta["sf"] == 7 and meta["cr"] == "4/8" and meta["expected"] == "deadbeef" and meta["times"] == 5 and meta["frequency_offset"] == 0
    assert sigmf.read_data(files[0] + ".sigmf-data").size > 5 * 30 * 1024


@pytest.mark.gpu
def test_qa_testsuite_on_device(tmp_path):
    """Reduced matrices (SF7-9 of short_rn incl. an LO offset of 1.5 kHz, SF7-8 of decode_long): every payload must come back."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test needs a GPU: the HIP path has no CPU fallback")
    gen, qa_mod = _load("generate_test_suites"), _load("qa_testsuite")
    suites = tmp_path / "test-suites"
    gen.generate(str(suites), gen.SHORT, sfs=[7, 8, 9], frequency_offset=1500)
    gen.generate(str(suites), gen.DECODE_LONG, sfs=[7, 8])
    qa = qa_mod.qa_testsuite(str(suites), port=40911)
    out = qa.run(print_output=False)
    qa.close()
    assert set(out) == {"short_rn", "decode_long"}
    assert out["short_rn"].num_total_payloads == 3 * 4 * 16 and out["short_rn"].num_total_correct_payloads == 3 * 4 * 16
    assert out["decode_long"].num_total_payloads == 2 and out["decode_long"].num_total_correct_payloads == 2
    md = open(tmp_path / "test-results" / "short_rn.md").read()
    assert "Total payloads passed: 192 out of 192 (100.00%)" in md and md.count("### 868.1 MHz, SF") == 12
