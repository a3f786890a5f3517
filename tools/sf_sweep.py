#!/usr/bin/env python3
"""BASELINE config 3: per-SF throughput (IQ Msamples/s, symbols/s) and walker-kernel GB/s vs the HBM roofline.
usage: tools/sf_sweep.py [packets_per_sf [first_sf [last_sf]]]   (prints one JSON line per SF; 0 packets = the reduced default set)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi
pk = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sf_lo = int(sys.argv[2]) if len(sys.argv) > 2 else 7
sf_hi = int(sys.argv[3]) if len(sys.argv) > 3 else 12
for sf in range(sf_lo, sf_hi + 1):
    n = pk or {7: 256, 8: 256, 9: 128, 10: 64, 11: 32, 12: 16}[sf]
    cfg, iq, offs, lens, expect = bench.make_workload(sf, 4, n, 32, min(8, n), seed=100 * sf + 4)
    d = torch.from_numpy(iq.view(np.float32)).cuda()
    h = capi.Handle(sf=sf, cr=4, reduced_rate=(sf > 10), demod=2)
    def step():
        h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)
        return h.drain()
    fr = step()
    got = {}
    for b, i in fr: got.setdefault(i.stream, []).append(b[15:])
    ok = all(got.get(s, []) == expect[s] for s in range(len(offs)))
    steps = 5 if sf < 11 else 2
    torch.cuda.synchronize(); t0 = time.perf_counter(); wk = 0.0
    for _ in range(steps):
        step(); wk += h.timing().walker_ms
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / steps
    print(json.dumps({"sf": sf, "packets": n, "items": int(iq.size), "bit_exact": ok, "Msamples_per_s": round(iq.size / el / 1e6, 1),
                      "symbols_per_s": round(iq.size / el / cfg.sps, 1), "ms_per_pass": round(el * 1e3, 3),
                      "walker_ms": round(wk / steps, 3), "kernel_GBps": round(8 * iq.size / (wk / steps * 1e-3) / 1e9, 1),
                      "frac_of_8TBps": round(8 * iq.size / (wk / steps * 1e-3) / 8e12, 5)}))
    h.close()
