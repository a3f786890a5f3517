#!/usr/bin/env python3
"""Per-state time of the walker for one SF (LORA_HIP_DEBUG accounting) + pass timing.  usage: tools/w3_profile.py sf [packets]"""
import os, sys, time
os.environ["LORA_HIP_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi
sf = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg, iq, offs, lens, expect = bench.make_workload(sf, 4, n, 32, min(8, n), seed=100 * sf + 4)
d = torch.from_numpy(iq.view(np.float32)).cuda()
h = capi.Handle(sf=sf, cr=4, reduced_rate=(sf > 10), demod=2)
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)
    fr = h.drain()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    tm = h.timing()
    print(f"pass {k}: {el*1e3:.3f} ms, walker {tm.walker_ms:.3f} ms, frames {len(fr)}, plan {h.plan()}", flush=True)
