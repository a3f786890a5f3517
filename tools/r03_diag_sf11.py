#!/usr/bin/env python3
"""which frames of the SF11 / SF12 gradient decode differ from the oracle's, and where their headers are (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, concurrent.futures as cf
import bench
from gr_lora_amd import capi
from oracle import oracle as O
for sf, cr in ((11, 1), (11, 4), (12, 1)):
    cfg, iq, offs, lens, expect = bench.make_workload(sf, cr, 256, 32, 8, seed=100 * sf + cr)
    kw = dict(sf=sf, cr=4, reduced_rate=True)
    def one(k):
        o = O.Oracle(demod=0, **kw); o.run(iq[offs[k]:offs[k] + lens[k]]); return o.frames(), o.frame_positions()
    with cf.ThreadPoolExecutor(8) as ex: want = list(ex.map(one, range(8)))
    d = torch.from_numpy(iq.view(np.float32)).cuda()
    h = capi.Handle(demod=0, **kw)
    h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)
    by = {}
    for g, i in h.drain(): by.setdefault(i.stream, []).append((g, i.header_pos))
    h.close()
    tot = mv = df = dfm = 0
    for s in range(8):
        gf = by.get(s, []); wf, wp = want[s]
        if len(gf) != len(wf): print("sf", sf, "cr", cr, "stream", s, "count", len(gf), len(wf)); continue
        for (g, gp), w, p in zip(gf, wf, wp):
            tot += 1; mv += gp != p; df += g != w; dfm += (g != w and gp != p)
            if g != w and gp == p: print("   same pos, differs:", sf, cr, s, gp, sum(a != b for a, b in zip(g, w)), "bytes")
    print("sf%d cr%d: %d frames, %d header positions differ, %d frames differ (%d of them with a moved header)" % (sf, cr, tot, mv, df, dfm), flush=True)
