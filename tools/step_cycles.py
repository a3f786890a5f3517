#!/usr/bin/env python3
"""Per-state latency breakdown of the walker (device clock per state-machine step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gr_lora_amd import capi, synth
names = ["DETECT", "SYNC", "FIND_SFD", "PAUSE", "HEADER", "PAYLOAD"]
for sf in [int(a) for a in sys.argv[1:]] or [7]:
    cfg = synth.TxConfig(sf=sf, cr=4, reduced_rate=(sf > 10))
    rng = np.random.default_rng(1)
    st = synth.build_stream([bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(2)], cfg, rng=rng)
    dev = torch.from_numpy(st.iq.view(np.float32)).cuda()
    for demod in (2, 0):
        h = capi.Handle(sf=sf, cr=4, reduced_rate=(sf > 10), demod=demod, flags=capi.FLAG_TRACE)
        for rep in range(2):
            h.trace_clear()
            h.decode_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size], 0)
        tr = h.trace()
        tot = {}
        for t in tr:
            tot.setdefault(t[0], []).append(t[7])
        print("sf", sf, "demod", demod, "frames", len(h.drain()), "walker_ms %.3f" % h.timing().walker_ms,
              {names[k]: (len(v), int(np.min(v)), int(np.mean(v))) for k, v in sorted(tot.items())})
        h.close()
