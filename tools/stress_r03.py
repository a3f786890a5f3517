#!/usr/bin/env python3
"""Determinism / correctness stress of the round-3 kernels: every (SF, demodulator) family, explicit and implicit header, repeated passes over the
bench workloads: the frames of every pass must be identical to the first pass's (and, FFT demodulators, to the payloads as sent), with no slow-path
re-launch.  usage: tools/stress_r03.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi, synth
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
total_bad = 0
for sf, packets in ((7, 1024), (8, 512), (9, 256), (10, 256), (11, 128), (12, 64)):
    cfg, iq, offs, lens, expect = bench.make_workload(sf, 4, packets, 32, 8, seed=100 * sf + 4)
    d = torch.from_numpy(iq.view(np.float32)).cuda()
    for demod in (0, 2):
        h = capi.Handle(sf=sf, cr=4, reduced_rate=(sf > 10), demod=demod)
        ref, bad = None, 0
        for it in range(iters):
            h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)
            fr = h.drain()
            tm = h.timing()
            key = [(b, i.stream, i.header_pos) for b, i in fr]
            if ref is None:
                ref = key
                got = {}
                for b, i in fr: got.setdefault(i.stream, []).append(b[15:])
                ok = all(got.get(s, []) == expect[s] for s in range(len(offs)))
                print("sf%d demod%d %s: %d frames, as sent: %s" % (sf, demod, h.kernel_name(), len(fr), ok), flush=True)
                if demod == 2 and not ok: bad += 1
            if key != ref or tm.slow_path_relaunches:
                bad += 1
                print("   iter", it, "frames", len(fr), "slow", tm.slow_path_relaunches, "same" if key == ref else "DIFFERENT", flush=True)
        h.close()
        total_bad += bad
    # implicit header, many short streams
    icfg = synth.TxConfig(sf=sf, cr=3, crc=False, implicit=True, reduced_rate=(sf > 10))
    rng = np.random.default_rng(sf)
    sts = [synth.build_stream([bytes(rng.integers(0, 256, int(rng.integers(4, 30)), dtype=np.uint8)) for _ in range(3)], icfg, rng=rng) for _ in range(16)]
    iiq = np.concatenate([s.iq for s in sts]); ioffs = np.cumsum([0] + [s.iq.size for s in sts[:-1]]).tolist(); ilens = [s.iq.size for s in sts]
    dd = torch.from_numpy(iiq.view(np.float32)).cuda()
    for demod in (0, 2):
        h = capi.Handle(sf=sf, cr=3, crc=False, implicit=True, reduced_rate=(sf > 10), demod=demod)
        ref, bad = None, 0
        for it in range(max(4, iters // 3)):
            h.decode_device(dd.data_ptr(), iiq.size, ioffs, ilens, 0)
            key = [(b, i.stream, i.header_pos) for b, i in h.drain()]
            if ref is None:
                ref = key
                print("sf%d demod%d implicit %s: %d frames of 48" % (sf, demod, h.kernel_name(), len(key)), flush=True)
                if len(key) != 48: bad += 1
            if key != ref: bad += 1; print("   implicit iter", it, "DIFFERENT", flush=True)
        h.close()
        total_bad += bad
print("bad:", total_bad)
