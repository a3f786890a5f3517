#!/usr/bin/env python3
"""GPU box: the reference's shipped mode (gradient demodulator) at BASELINE config 3's full size against the fixtures the compiled
reference made (tests/golden/fullsize_ref.json) - header positions and frames that differ, per cell, with strict SYNC (default) and
with LORA_HIP_FLAG_FAST_SYNC (the closed-form maximum alone, the behaviour up to round 3).  Output -> profiles/r04_strict_sync_diag.txt"""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi

FIX = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fullsize_ref.json")))
cells = sys.argv[1:] or ["config3-sf%d-cr%d" % (sf, cr) for sf in (7, 9, 10, 11, 12) for cr in (1, 4)]
for tag in cells:
    fx = FIX[tag]
    cfg, iq, offs, lens, expect = bench.make_workload(fx["sf"], fx["cr"], fx["packets"], fx["payload"], fx["streams"], seed=fx["seed"])
    d = torch.from_numpy(iq.view(np.float32)).cuda()
    for name, flags in (("strict", 0), ("fast_sync", capi.FLAG_FAST_SYNC)):
        h = capi.Handle(demod=0, flags=flags, **fx["decoder_kw"])
        h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)          # warm
        h.drain()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)
        t1 = time.perf_counter()
        tm = h.timing()
        ms = tm.walker_ms
        by = {}
        for g, i in h.drain(): by.setdefault(i.stream, []).append((g, i.header_pos))
        h.close()
        tot = mv = df = miss = 0
        for s, want in enumerate(fx["per_stream"]):
            gf = by.get(s, [])
            if len(gf) != want["frames"]: miss += abs(len(gf) - want["frames"]); continue
            for (g, gp), p, sha in zip(gf, want["header_pos"], want["frame_sha"]):
                tot += 1; mv += gp != p; df += hashlib.sha256(g).hexdigest()[:10] != sha
        print("%-18s %-9s frames %4d  header positions differing %4d  frames differing %4d  missing %d  walker %.3f ms in %d launch(es), %d jobs, %d probes, %d serial re-runs (call %.3f ms)"
              % (tag, name, tot, mv, df, miss, ms, tm.walker_launches, tm.jobs, tm.probes, tm.slow_path_relaunches, 1e3 * (t1 - t0)), flush=True)
