#!/usr/bin/env python3
"""Throughput of the symbol demodulator alone (lora_hip_demod_symbols_ex_device): usage tools/demod_bench.py sf[,sf...] [n_symbols]
DEMOD_BENCH_RATE=5e5 / 2.5e5: decimation 4 / 2; DEMOD_BENCH_MODE=0: the gradient estimator"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gr_lora_amd import capi, synth
for sf in [int(a) for a in sys.argv[1].split(",")]:
    rate, mode = float(os.environ.get("DEMOD_BENCH_RATE", "1e6")), int(os.environ.get("DEMOD_BENCH_MODE", "2"))
    cfg = synth.TxConfig(sf=sf, samp_rate=rate)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else max(2048, (1 << 27) // cfg.sps)
    rng = np.random.default_rng(sf)
    up = synth.base_upchirp(cfg)
    base = np.concatenate([np.roll(up, -int(s) * cfg.decim) for s in rng.integers(0, cfg.nbins, 64)]).astype(np.complex64)
    iq = np.tile(base, n // 64)
    offs = np.arange(n, dtype=np.int64) * cfg.sps
    d = torch.from_numpy(iq.view(np.float32)).cuda()
    h = capi.Handle(sf=sf, demod=mode, samp_rate=rate)
    g, f = h.demod_symbols_ex_device(d.data_ptr(), iq.size, offs, mode)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.demod_symbols_ex_device(d.data_ptr(), iq.size, offs, mode)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"sf{sf} D{cfg.decim} mode {mode}: {n} symbols, {iq.size / best / 1e9:.1f} Gsamples/s ({8 * iq.size / best / 8e12 * 100:.1f}% of 8 TB/s), {best * 1e3:.3f} ms incl. host", flush=True)
    h.close()
