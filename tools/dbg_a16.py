#!/usr/bin/env python3
"""GPU box: one case of tests/test_gpu_a16.py::test_disable_drift_correction - where frames / traces leave the oracle's.
usage: dbg_a16.py SF DEMOD"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gr_lora_amd import capi, synth
from oracle import oracle as O
sf, demod = int(sys.argv[1]), int(sys.argv[2])
n = {7: 6, 8: 5, 9: 4, 10: 3, 11: 2, 12: 1}[sf]
for cr in ((4, 1) if sf < 11 else (4,)):
    cfg = synth.TxConfig(sf=sf, cr=cr, reduced_rate=(sf > 10))
    rng = np.random.default_rng(31 * sf + cr)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(4, 28)), dtype=np.uint8)) for _ in range(n)]
    noisy = synth.build_stream(payloads, cfg, rng=rng, noise_sigma=synth.awgn_sigma_for_snr(42.0, cfg))
    clean = synth.build_stream(payloads, cfg, rng=np.random.default_rng(5 * sf + cr))
    kw = dict(sf=sf, cr=cr, reduced_rate=(sf > 10), disable_drift_correction=True)
    for name, st in (("noisy", noisy), ("clean", clean)):
        o = O.Oracle(demod=demod, **kw); o.enable_trace(); o.run(st.iq)
        dev = torch.from_numpy(st.iq.view(np.float32)).cuda()
        for flags in (0, capi.FLAG_FAST_SYNC):
            h = capi.Handle(demod=demod, flags=capi.FLAG_TRACE | flags, **kw)
            h.decode_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size], 0)
            got = h.drain(); tr = h.trace(); h.close()
            ot = o.trace()
            gf, of = [g.hex() for g, _ in got], [f.hex() for f in o.frames()]
            print("cr", cr, name, "flags", flags, "frames equal", gf == of, "steps", len(tr), len(ot), "pos", [i.header_pos for _, i in got], o.frame_positions())
            if gf != of:
                for a, b in zip(gf, of):
                    if a != b: print("   frame dev", a[30:], "\n   frame ora", b[30:])
            nd = 0
            for i, (a, b) in enumerate(zip(tr, ot)):
                if tuple(a[:5]) != tuple(b[:5]):
                    z = bool(np.any(st.iq[b[1]:b[1] + (8 << sf) + 1] == 0))
                    print("   step", i, "device", tuple(a[:6]), "oracle", tuple(b[:6]), "zero-in-window", z)
                    nd += 1
                    if nd > 6: break
