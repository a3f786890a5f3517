import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi
cfg, iq, offs, lens, expect = bench.make_workload(7, 4, 256, 32, 2, 2)
d = torch.from_numpy(iq.view(np.float32)).cuda()
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
h = capi.Handle(sf=7, cr=4, demod=2, segment_symbols=seg)
ref = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)
    fr = h.drain()
    key = {(i.stream, i.header_pos): b for b, i in fr}
    if ref is None:
        ref = key; continue
    if key != ref:
        miss = sorted(set(ref) - set(key)); extra = sorted(set(key) - set(ref))
        diff = [k for k in key if k in ref and key[k] != ref[k]]
        print("iter", it, "missing", miss[:4], "extra", extra[:4], "bytes differ", len(diff))
        for k in diff[:2]:
            a, b = ref[k], key[k]
            idx = [i for i in range(min(len(a), len(b))) if a[i] != b[i]]
            print("   ", k, "len", len(a), len(b), "diff idx", idx[:10], [hex(a[i]) for i in idx[:6]], [hex(b[i]) for i in idx[:6]])
