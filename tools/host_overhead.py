import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi
cfg, iq, offs, lens, expect = bench.make_workload(7, 4, 1024, 32, 8, 2)
d = torch.from_numpy(iq.view(np.float32)).cuda()
h = capi.Handle(sf=7, cr=4, demod=2)
for i in range(4):
    t0 = time.perf_counter(); h.decode_device(d.data_ptr(), iq.size, offs, lens, 0); t1 = time.perf_counter()
    fr = h.drain(); t2 = time.perf_counter()
    print("decode %.3f ms  drain %.3f ms  frames %d  walker %.3f" % ((t1-t0)*1e3, (t2-t1)*1e3, len(fr), h.timing().walker_ms))
