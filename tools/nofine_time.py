#!/usr/bin/env python3
"""Diagnostic: walker time per pass with and without the per-symbol fine_sync (disable_drift_correction): what the window's
instantaneous frequency costs inside the walker.  usage: tools/nofine_time.py sf [packets]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi
sf = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg, iq, offs, lens, expect = bench.make_workload(sf, 4, n, 32, min(8, n), seed=2 if sf == 7 and n == 1024 else 100 * sf + 4)
d = torch.from_numpy(iq.view(np.float32)).cuda()
for nodrift in (False, True):
    h = capi.Handle(sf=sf, cr=4, reduced_rate=(sf > 10), demod=2, disable_drift_correction=nodrift)
    ts = []
    for k in range(12):
        h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)
        fr = h.drain()
        ts.append(h.timing().walker_ms)
    print("sf %d disable_drift_correction=%s: walker ms min %.4f median %.4f, frames %d" % (sf, nodrift, min(ts), float(np.median(ts)), len(fr)))
    h.close()
