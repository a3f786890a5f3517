#!/usr/bin/env python3
"""GPU box: tests/test_gpu_a16.py::test_disable_drift_correction[0-12] (clean) - where the device trace leaves the oracle's, and what the
gradient estimator sees in that window (float64 bin averages: the largest drops)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gr_lora_amd import capi, synth
from oracle import oracle as O
sf, cr, n = 12, 4, 1
cfg = synth.TxConfig(sf=sf, cr=cr, reduced_rate=True)
rng = np.random.default_rng(31 * sf + cr)
payloads = [bytes(rng.integers(0, 256, int(rng.integers(4, 28)), dtype=np.uint8)) for _ in range(n)]
st = synth.build_stream(payloads, cfg, rng=np.random.default_rng(5 * sf + cr))
kw = dict(sf=sf, cr=cr, reduced_rate=True, disable_drift_correction=True)
o = O.Oracle(demod=0, **kw); o.enable_trace(); o.run(st.iq)
dev = torch.from_numpy(st.iq.view(np.float32)).cuda()
for flags in (0, capi.FLAG_FAST_SYNC):
    h = capi.Handle(demod=0, flags=capi.FLAG_TRACE | flags, **kw)
    h.decode_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size], 0)
    got = h.drain(); tr = h.trace(); h.close()
    ot = o.trace()
    print("flags", flags, "frames equal", [g.hex() for g, _ in got] == [f.hex() for f in o.frames()], "steps", len(tr), len(ot))
    for i, (a, b) in enumerate(zip(tr, ot)):
        if tuple(a[:5]) != tuple(b[:5]):
            print("  step", i, "device", a[:6], "oracle", b[:6])
            pos = b[1]; sps = 8 << sf
            x = st.iq[pos:pos + sps + 1].astype(np.complex128)
            f = np.angle(x[1:] * np.conj(x[:-1])); f[-1] = f[-2]
            avg = f.reshape(-1, 8).mean(axis=1)
            g = avg[:-1] - avg[1:]
            top = np.argsort(-g)[:4]
            print("   largest drops (i, gradient):", [(int(t) + 1, float(g[t])) for t in top], " -> bins", [int(((1 << sf) - (t + 2)) % (1 << sf)) for t in top])
            f32 = O.instantaneous_frequency(st.iq[pos:pos + sps])
            a32 = np.array([np.float32(sum(np.float32(v) for v in f32[8 * k:8 * k + 8])) / np.float32(8) for k in (top.tolist() + (top + 1).tolist())])
            print("   oracle-order float32 averages at those bins:", a32.tolist())
