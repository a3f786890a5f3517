"""SigMF reading / writing with the reference's meta schema
(apps/lora_receive_file_nogui.py:75-94, apps/generate_test_suites.py:39-46,68-82)."""
from __future__ import annotations

import json
import os
from typing import Dict, Tuple

import numpy as np


def write_trace(path_base: str, iq: np.ndarray, sample_rate: float, capture_freq: float, transmit_freq: float, sf: int,
                cr: str, bw: int, prlen: int, crc: bool, implicit: bool, expected_hex: str, times: int,
                hw: str = "synthetic", frequency_offset: float = 0) -> Tuple[str, str]:
    """`frequency_offset` is the capture's LO calibration offset (generate_test_suites.py -F, default 0): the harness
    removes it with an extra translating filter in front of the receiver (qa_testsuite.py:233)."""
    data_path, meta_path = path_base + ".sigmf-data", path_base + ".sigmf-meta"
    np.ascontiguousarray(iq, dtype=np.complex64).tofile(data_path)
    meta = {
        "global": {"core:datatype": "cf32_le", "core:version": "0.0.1", "core:sample_rate": sample_rate,
                   "core:hw": hw, "core:description": "synthetic LoRa capture (gr_lora_amd.synth)"},
        "captures": [{"core:sample_start": 0, "core:frequency": capture_freq,
                      "lora:frequency": transmit_freq, "lora:frequency_offset": frequency_offset,
                      "lora:sf": sf, "lora:cr": cr, "lora:bw": bw, "lora:prlen": prlen, "lora:crc": crc,
                      "lora:implicit": implicit, "test:expected": expected_hex, "test:times": times}],
        "annotations": [],
    }
    with open(meta_path, "w") as f:
        json.dump(meta, f, indent=2)
    return data_path, meta_path


def read_meta(meta_path: str) -> Dict:
    meta = json.load(open(meta_path))
    g, c = meta["global"], meta["captures"][0]
    return {"sample_rate": g["core:sample_rate"], "capture_freq": c["core:frequency"],
            "transmit_freq": c["lora:frequency"], "sf": c["lora:sf"], "cr": c["lora:cr"], "bw": int(c["lora:bw"]),
            "prlen": c["lora:prlen"], "crc": c["lora:crc"], "implicit": c["lora:implicit"],
            "expected": c["test:expected"], "times": c["test:times"], "frequency_offset": c.get("lora:frequency_offset", 0)}


def read_data(data_path: str) -> np.ndarray:
    return np.fromfile(data_path, dtype=np.complex64)


class LoRaConfig:
    """python/loraconfig.py: cr given as "4/8" -> cr_num."""

    def __init__(self, freq, sf, cr, bw=125e3, prlen=8, crc=True, implicit=False):
        self.freq, self.sf, self.cr, self.bw, self.prlen, self.crc, self.implicit = freq, sf, cr, bw, prlen, crc, implicit
        self.cr_num = int(str(cr).split("/")[1]) - 4 if isinstance(cr, str) else int(cr)

    def string_repr(self):
        return "%.1f MHz, SF %d, CR %s, BW %d kHz, prlen %d, crc %s, implicit %s" % (
            self.freq / 1e6, self.sf, self.cr, int(self.bw / 1e3), self.prlen, "on" if self.crc else "off",
            "on" if self.implicit else "off")
