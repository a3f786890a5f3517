"""Where a bench step's wall time goes on the host side (per call, after warm-up)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi, gather
cfg, iq, offs, lens, expect = bench.make_workload(7, 4, 1024, 32, 8, 2)
d = torch.from_numpy(iq.view(np.float32)).cuda()
dev = torch.device("cuda", 0)
h = capi.Handle(sf=7, cr=4, demod=2)
acc = np.zeros(4)
N = 12
for i in range(N + 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); h.decode_device(d.data_ptr(), iq.size, offs, lens, 0); t1 = time.perf_counter()
    mine = h.drain_slots(gather.SLOT_BYTES); t2 = time.perf_counter()
    slots, counts = gather.gather_slots(mine, dev); t3 = time.perf_counter()
    if i >= 3:
        acc += [t1 - t0, t2 - t1, t3 - t2, h.timing().walker_ms * 1e-3]
acc *= 1e3 / N
print("decode_device %.3f ms (walker kernels %.3f ms)  drain_slots %.3f ms  gather_slots %.3f ms  frames %d" % (acc[0], acc[3], acc[1], acc[2], mine.shape[0]))
