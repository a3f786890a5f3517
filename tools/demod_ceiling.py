#!/usr/bin/env python3
"""The standalone symbol demodulators streaming over resident IQ: every window independent, no state machine, no acquisition - the rate
the decode rounds of a walker cannot exceed.  One launch of lora_hip_demod_symbols_ex_device over ~1e8 items per
spreading factor and demodulator; run it under `rocprofv3 --kernel-trace --stats` and feed the kernel-stats CSV to --collect to write
profiles/r04_demod_ceiling.json (bench.py quotes it as roofline.valu_ceiling_frac).
usage: demod_ceiling.py run [sf,sf,...]      |      demod_ceiling.py --collect <kernel_stats.csv> <out.json>"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ITEMS = 1 << 27

def plan(sf):
    sps = 8 << sf
    return max(2048, ITEMS // sps)

if sys.argv[1] == "run":
    import numpy as np, torch
    from gr_lora_amd import capi, synth
    for sf in [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "7,8,9,10,11,12").split(",")]:
        cfg = synth.TxConfig(sf=sf)
        n = plan(sf)
        rng = np.random.default_rng(sf)
        up = synth.base_upchirp(cfg)
        base = np.concatenate([np.roll(up, -int(s) * 8) for s in rng.integers(0, cfg.nbins, 64)]).astype(np.complex64)
        iq = np.concatenate([np.zeros(cfg.sps, np.complex64), np.tile(base, n // 64), np.zeros(cfg.sps, np.complex64)])  # (a symbol of margin at both ends)
        offs = (1 + np.arange(n, dtype=np.int64)) * cfg.sps
        d = torch.from_numpy(iq.view(np.float32)).cuda()
        for demod in (2, 0):
            h = capi.Handle(sf=sf, demod=demod)
            for _ in range(4):
                h.demod_symbols_ex_device(d.data_ptr(), iq.size, offs, demod)
            h.close()
        del d
        torch.cuda.empty_cache()
else:
    import bench
    rows = list(csv.DictReader(open(sys.argv[2])))
    cells = {}
    for r in rows:
        name = r.get("Name") or r.get("KernelName") or ""
        if "demod_symbols" not in name:
            continue
        avg_ns = float(r.get("AverageNs") or r.get("Average") or 0)
        for sf in range(7, 13):
            for demod, tag in ((2, ""), (0, "grad")):
                if demod == 0:
                    pats = ["demod_symbols_wave_grad_kernel<%d>" % sf if sf <= 8 else "demod_symbols_w3_grad_kernel<%d>" % sf]
                else:
                    pats = ["demod_symbols_wave_kernel<%d," % sf] if sf <= 9 else ["demod_symbols_w3_kernel<%d, false" % sf, "demod_symbols_w3_kernel<%d,false" % sf, "demod_symbols_team_kernel<%d>" % sf]
                if any(p in name for p in pats):
                    items = plan(sf) * (8 << sf)
                    cells["sf%d-demod%d" % (sf, demod)] = {"kernel": name, "calls": int(r.get("Calls") or 0), "avg_ms": round(avg_ns / 1e6, 4), "items": items,
                                                           "GBps": round(8.0 * items / (avg_ns * 1e-9) / 1e9, 1),
                                                           "frac_of_hbm_peak": round(8.0 * items / (avg_ns * 1e-9) / 1e9 / bench.HBM_PEAK_GBS, 5)}
    json.dump({"what": "standalone symbol demodulators (lora_hip_demod_symbols_ex_device) streaming over resident IQ, rocprofv3 --kernel-trace --stats averages",
               "source_hash": bench.source_hash(), "cells": cells}, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(cells, indent=1))
