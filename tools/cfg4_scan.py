#!/usr/bin/env python3
"""BASELINE config 4 (8 continuous SF9 channels per GPU): walker time per pass against the segment length.
usage: tools/cfg4_scan.py [seconds] [segment_symbols ...]   (0 = the automatic plan)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
segs = [int(a) for a in sys.argv[2:]] or [0, 40, 48, 56, 64, 80, 100, 128, 160]
cfg, iq, offs, lens, expect = bench.make_gateway_workload(list(range(8)), seconds, 9)
d = torch.from_numpy(iq.view(np.float32)).cuda()
want = sum(len(e) for e in expect)
for seg in segs:
    h = capi.Handle(sf=9, cr=4, demod=2, segment_symbols=seg)
    ts = []
    for k in range(6):
        h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)
        fr = h.drain()
        ts.append(h.timing().walker_ms)
    print("segment_symbols %3d: plan %s, walker ms min %.4f median %.4f, frames %d of %d" % (seg, h.plan(), min(ts), float(np.median(ts)), len(fr), want), flush=True)
    h.close()
