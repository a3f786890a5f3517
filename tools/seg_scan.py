import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi
cfg, iq, offs, lens, expect = bench.make_workload(7, 4, 1024, 32, 8, 2)
d = torch.from_numpy(iq.view(np.float32)).cuda()
for seg in [int(a) for a in sys.argv[1:]] or [0, 64, 90, 110, 140, 180, 226, 300]:
    h = capi.Handle(sf=7, cr=4, demod=2, segment_symbols=seg)
    best = 1e9; wk = 0
    for i in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); h.decode_device(d.data_ptr(), iq.size, offs, lens, 0); t1 = time.perf_counter()
        fr = h.drain()
        if i >= 2 and t1 - t0 < best: best = t1 - t0; tm = h.timing(); wk = tm.walker_ms
    print("seg %4d: decode %.3f ms walker %.3f ms jobs %d probes %d slow %d frames %d" % (seg, best * 1e3, wk, tm.jobs, tm.probes, tm.slow_path_relaunches, len(fr)))
    h.close()
