import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from gr_lora_amd import capi, synth
from oracle import oracle as O
import test_gpu_zeros as Z
sf = int(sys.argv[1]); demod = int(sys.argv[2]); trace = int(sys.argv[3])
iq = Z._planted(sf, 4, 11 * sf + demod, 5, 0.4 / (8 << sf), 6)
dev = torch.from_numpy(iq.view(np.float32)).cuda()
h = capi.Handle(demod=demod, sf=sf, cr=4, flags=(capi.FLAG_TRACE if trace else 0))
print("kernel", h.kernel_name(), flush=True)
h.decode_device(dev.data_ptr(), iq.size, [0], [iq.size], 0)
got = h.drain()
print("frames", len(got), flush=True)
o = O.Oracle(demod=demod, sf=sf, cr=4); o.run(iq)
print("oracle frames", len(o.frames()), [g.hex() for g, _ in got] == [f.hex() for f in o.frames()], flush=True)
