"""Per-job cycle statistics (LORA_HIP_DEBUG=1 output of the runtime) for a given segment length on the bench workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi
cfg, iq, offs, lens, expect = bench.make_workload(7, 4, 1024, 32, 8, 2)
d = torch.from_numpy(iq.view(np.float32)).cuda()
for seg in [int(a) for a in sys.argv[1:]]:
    h = capi.Handle(sf=7, cr=4, demod=2, segment_symbols=seg)
    for i in range(3):
        sys.stderr.write("=== seg %d pass %d\n" % (seg, i)); sys.stderr.flush()
        h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)
        h.drain()
    h.close()
