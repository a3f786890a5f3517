#!/usr/bin/env python3
"""Generates the committed golden fixtures.

The reference itself cannot run in the build container (GNU Radio / liquid-dsp /
VOLK absent), so the pinned answers are:
  * the README known answer (README.md:75-85) and the SURVEY Appendix-C symbol list,
  * outputs of the CPU oracle (oracle/, pinned against the two above) on seeded
    synthetic streams -- frames, header positions and per-symbol bins.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gr_lora_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    cases = []
    # one small IQ file: SF7 CR4/8, the README packet twice
    cfg = synth.TxConfig(sf=7, cr=4, crc=True, sync_shifts=(24, 32))
    st = synth.build_stream([bytes.fromhex("deadbeef")] * 2, cfg, gaps=[3000, 2500], tail_symbols=2.5)
    st.iq.tofile(os.path.join(HERE, "sf7_cr4_deadbeef_x2.cf32"))
    for sf in (7, 8, 9, 10):
        for cr in (1, 2, 3, 4):
            seed = 100 * sf + cr
            rng = np.random.default_rng(seed)
            cfg = synth.TxConfig(sf=sf, cr=cr, crc=True)
            payloads = [bytes(rng.integers(0, 256, int(rng.integers(1, 24)), dtype=np.uint8)) for _ in range(3)]
            gaps = [int(g) for g in rng.integers(2 * cfg.sps, 6 * cfg.sps, 3)]
            s = synth.build_stream(payloads, cfg, gaps=gaps)
            entry = {"sf": sf, "cr": cr, "seed": seed, "payloads": [p.hex() for p in payloads], "gaps": gaps,
                     "shifts": [[h, q] for h, q in s.shifts], "n_items": int(s.iq.size), "modes": {}}
            for mode in (0, 1, 2):
                o = O.Oracle(sf=sf, cr=4, crc=True, demod=mode)
                o.enable_trace()
                o.run(s.iq)
                bins = [t[3] for t in o.trace() if t[0] in (4, 5)]
                entry["modes"][str(mode)] = {"frames": [f.hex() for f in o.frames()], "header_pos": o.frame_positions(), "bins": bins}
            cases.append(entry)
    golden = {
        "readme_known_answer": "049040deadbeef700d",
        "appendix_c": {"header": [29, 1, 97, 125, 37, 109, 1, 97],
                       "payload": [119, 51, 20, 1, 22, 82, 37, 58, 2, 17, 28, 115, 117, 98, 110, 7]},
        "iq_file": {"name": "sf7_cr4_deadbeef_x2.cf32", "sf": 7, "cr": 4, "gaps": [3000, 2500],
                    "frames": ["00" * 15 + "049040deadbeef700d"] * 2},
        "cases": cases,
    }
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(golden, f)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
