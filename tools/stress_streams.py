import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg, iq, offs, lens, expect = bench.make_workload(7, 4, n, 32, n, 2)
d = torch.from_numpy(iq.view(np.float32)).cuda()
h = capi.Handle(sf=7, cr=4, demod=2, segment_symbols=100000)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)
    fr = h.drain()
    got = {}
    for b, i in fr: got.setdefault(i.stream, []).append(b[15:])
    wrong = [s for s in range(n) if got.get(s, []) != expect[s]]
    if wrong:
        bad += 1
        s = wrong[0]
        g = got.get(s, [b""])[0] if got.get(s) else b""
        e = expect[s][0]
        print("iter", it, "wrong streams", len(wrong), wrong[:6], "first: len", len(g), len(e), "diffpos", [i for i in range(min(len(g), len(e))) if g[i] != e[i]][:12])
print("bad", bad)
