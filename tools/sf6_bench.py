#!/usr/bin/env python3
"""SF6 (implicit header) walker throughput, fast family against the generic kernels: usage tools/sf6_bench.py [decim] [streams] (LORA_HIP_NO_FAST=1 for the generic ones).
An implicit-header stream is ONE job (the payload length is not known ahead: no speculation segments), so the device fills up with streams, not with packets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gr_lora_amd import capi, synth
decim = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 512
kw = dict(sf=6, cr=4, crc=False, implicit=True, samp_rate=125000.0 * decim)
cfg = synth.TxConfig(**kw)
rng = np.random.default_rng(6)
pieces, offs, lens = [], [], []
off = 0
for s in range(n_streams):
    st = synth.build_stream([bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(1024 // n_streams)], cfg, rng=rng, gap_symbols=(2.0, 6.0), noise_sigma=synth.awgn_sigma_for_snr(50.0, cfg))
    pieces.append(st.iq); offs.append(off); lens.append(st.iq.size); off += st.iq.size
iq = np.concatenate(pieces)
d = torch.from_numpy(iq.view(np.float32)).cuda()
for demod in (2, 0):
    h = capi.Handle(demod=demod, **kw)
    ms, frames = [], 0
    for it in range(12):
        h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)
        frames = len(h.drain())
        ms.append(h.timing().walker_ms)
    k = float(np.median(ms[2:]))
    print("sf6 D%d demod %d %s: %d streams, %d items, %d frames, walker %.4f ms = %.1f Gsamples/s (%.1f %% of 8 TB/s)" % (decim, demod, h.kernel_name(), n_streams, iq.size, frames, k, iq.size / k / 1e6, 8 * iq.size / (k * 1e-3) / 8e12 * 100), flush=True)
    h.close()
