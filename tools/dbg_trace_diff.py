#!/usr/bin/env python3
"""Diagnostics: first step where the device trace leaves the oracle's, for a few named cases (run on the GPU box)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gr_lora_amd import capi, synth
from oracle import oracle as O

ST = O.ST_NAMES

def diff(name, iq, demod, seg=0, **kw):
    o = O.Oracle(demod=demod, **kw); o.enable_trace(); o.run(iq)
    h = capi.Handle(demod=demod, flags=capi.FLAG_TRACE, segment_symbols=seg, **kw)
    dev = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).cuda()
    h.decode_device(dev.data_ptr(), iq.size, [0], [iq.size], 0)
    got = h.drain(); tr = h.trace(); h.close()
    otr = o.trace()
    okf = [g for g, _ in got] == o.frames()
    print(f"== {name}: frames equal {okf} ({len(got)} vs {len(o.frames())}); steps {len(tr)} vs {len(otr)}")
    for i, (a, b) in enumerate(zip(tr, otr)):
        if tuple(a[:5]) != tuple(b[:5]) or (np.isfinite(b[5]) and abs(a[5] - b[5]) > 1e-3 * max(1, abs(b[5]))):
            for j in range(max(0, i - 3), min(len(tr), len(otr), i + 3)):
                print("   ", j, "GPU", ST[tr[j][0]], tr[j][1:6], "| ORA", ST[otr[j][0]], otr[j][1:6], "<--" if j == i else "")
            break
    else:
        print("    traces identical over the common prefix")

which = sys.argv[1:] or ["nodrift8", "impl11", "noisy11", "golden"]
if "nodrift8" in which:
    sf, cr = 8, 4
    cfg = synth.TxConfig(sf=sf, cr=cr); rng = np.random.default_rng(31 * sf + cr)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(4, 28)), dtype=np.uint8)) for _ in range(5)]
    st = synth.build_stream(payloads, cfg, rng=rng, noise_sigma=synth.awgn_sigma_for_snr(42.0, cfg))
    diff("nodrift sf8 demod1", st.iq, 1, sf=sf, cr=cr, disable_drift_correction=True)
if "impl11" in which:
    sf, cr, crc = 11, 3, True
    cfg = synth.TxConfig(sf=sf, cr=cr, crc=crc, reduced_rate=True, implicit=True); rng = np.random.default_rng(17 * sf + cr)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(6, 24)), dtype=np.uint8)) for _ in range(2)]
    st = synth.build_stream(payloads, cfg, rng=rng)
    for demod in (0, 2):
        diff(f"implicit sf11 demod{demod}", st.iq, demod, sf=sf, cr=cr, crc=crc, reduced_rate=True, implicit=True)
if "noisy11" in which:
    sf = 11
    rng = np.random.default_rng(900 + sf)
    pieces = []
    for i in range(4):
        cfg = synth.TxConfig(sf=sf, cr=int(rng.integers(1, 5)), reduced_rate=True)
        p = bytes(rng.integers(0, 256, int(rng.integers(3, 20)), dtype=np.uint8))
        pieces.append(synth.build_stream([p], cfg, rng=rng, tail_symbols=0.0).iq)
    sps = 8 << sf
    iq = np.concatenate(pieces + [np.zeros(3 * sps, np.complex64)])
    sigma = synth.awgn_sigma_for_snr(40.0, synth.TxConfig(sf=sf))
    iq = (iq + (rng.standard_normal(iq.size) + 1j * rng.standard_normal(iq.size)).astype(np.complex64) * np.float32(sigma / np.sqrt(2))).astype(np.complex64)
    diff("noisy sf11 demod2", iq, 2, sf=sf, cr=4, reduced_rate=True)
if "golden" in which:
    import json
    G = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "golden.json")))
    for case in G["cases"]:
        if case["sf"] < 11: continue
        cfg = synth.TxConfig(sf=case["sf"], cr=case["cr"], crc=case["crc"], reduced_rate=case["reduced_rate"], implicit=case["implicit"])
        st = synth.build_stream([bytes.fromhex(p) for p in case["payloads"]], cfg, gaps=case["gaps"])
        kw = dict(sf=case["sf"], cr=case["cr"], crc=case["crc"], implicit=case["implicit"], reduced_rate=case["reduced_rate"], disable_drift_correction=case["disable_drift_correction"])
        for demod in (0, 1, 2):
            diff(f"golden sf{case['sf']} cr{case['cr']} demod{demod}", st.iq, demod, **kw)
