"""Run-to-run determinism of a pass: the same frames at the same header positions, no slow-path relaunch, over many passes of one workload.
usage: tools/stress_determinism.py [iterations] [sf] [packets]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi
SF = int(sys.argv[2]) if len(sys.argv) > 2 else 7
PK = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
cfg, iq, offs, lens, expect = bench.make_workload(SF, 4, PK, 32, 8, 2 if (SF, PK) == (7, 1024) else 100 * SF + 4)
d = torch.from_numpy(iq.view(np.float32)).cuda()
h = capi.Handle(sf=SF, cr=4, demod=2)
ref = None
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)
    fr = h.drain()
    tm = h.timing()
    key = [(b, i.stream, i.header_pos) for b, i in fr]
    if ref is None:
        ref = key
        got = {}
        for b, i in fr: got.setdefault(i.stream, []).append(b[15:])
        print("first run ok vs expected:", all(got.get(s, []) == expect[s] for s in range(len(offs))), len(fr))
    if key != ref or tm.slow_path_relaunches:
        bad += 1
        print("iter", it, "frames", len(fr), "slow", tm.slow_path_relaunches, "walker_ms %.3f" % tm.walker_ms, "same" if key == ref else "DIFFERENT")
print("sf", SF, "packets", PK, "kernel", h.kernel_name(), "bad iterations:", bad)
