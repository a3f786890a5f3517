#!/usr/bin/env python3
"""Full-size fixtures made BY THE REFERENCE ITSELF.

BASELINE.json's configurations 2 and 3 at their full sizes - bench.make_workload rebuilds the IQ from the seeds at test
time - are run through oracle/_ref/libref_decoder.so, the parity build of
/root/reference/lib/decoder_impl.cc (unmodified, sequential VOLK stand-in), in its shipped configuration (gradient
demodulator, decoder_impl.cc:499), one decoder per stream.  Recorded per stream: the number of frames, sha256 over the
published frames (15 B loratap | 3 B PHY header | payload, back to back) and the sample position of every frame's first
header symbol.  tests/test_gpu_fullsize.py holds the GPU's DEMOD_GRAD path (walker2 / walker3 gradient kernels) and the
oracle to these.

Needs /root/reference (the build container).  Run from the repo root:  python tests/golden/make_fullsize_golden.py
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import ref as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# (tag, sf, cr, packets, payload, streams, seed) - the same calls tests/test_gpu_fullsize.py makes
CASES = [("config2-1stream", 7, 4, 1024, 32, 1, 2), ("config2-8streams", 7, 4, 1024, 32, 8, 2)]
for _sf in (7, 8, 9, 10, 11, 12):
    for _cr in (1, 2, 3, 4):   # BASELINE config 3: CR 4/5 - 4/8
        CASES.append(("config3-sf%d-cr%d" % (_sf, _cr), _sf, _cr, 256, 32, 8, 100 * _sf + _cr))


# the default bench workload as the ranks 1 .. 7 of a multi-GPU run synthesise it (bench.py: seed 2 + 1000 rank): `bench.py --demod 0 --gpus N` then has a yardstick on every rank
for _r in range(1, 8):
    CASES.append(("config2-8streams-rank%d" % _r, 7, 4, 1024, 32, 8, 2 + 1000 * _r))
CASES.append(("config3-sf8-cr4-1024packets", 8, 4, 1024, 32, 8, 804))  # the SF8 profile workload (tools/profile_all.sh: 1024 packets fill the device)


def digest(frames):
    h = hashlib.sha256()
    for f in frames:
        h.update(len(f).to_bytes(4, "little"))
        h.update(f)
    return h.hexdigest()


_CASE = {}


def _one(k):
    """one reference decoder over stream k, in a forked child (the reference's console capture swaps std::cout's buffer: its
    instances cannot run on threads of one process)"""
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)                                    # the reference prints every frame (decoder_impl.cc:832,:872)
    iq, offs, lens, kw = _CASE["iq"], _CASE["offs"], _CASE["lens"], _CASE["kw"]
    r = R.Reference(**kw)
    r.run(iq[offs[k]:offs[k] + lens[k]])
    return r.frames(), r.frame_positions()


def run_case(tag, sf, cr, packets, payload, streams, seed):
    import multiprocessing as mp
    cfg, iq, offs, lens, expect = bench.make_workload(sf, cr, packets, payload, streams, seed=seed)
    kw = dict(sf=sf, cr=4, crc=True, reduced_rate=(sf > 10))
    _CASE.update(iq=iq, offs=offs, lens=lens, kw=kw)
    t0 = time.time()
    sys.stdout.flush()
    with mp.get_context("fork").Pool(min(8, len(offs))) as pool:   # (children see the IQ copy-on-write)
        res = pool.map(_one, range(len(offs)))
    entry = {"tag": tag, "sf": sf, "cr": cr, "packets": packets, "payload": payload, "streams": streams, "seed": seed, "decoder_kw": kw,
             "n_items": int(iq.size), "source": "reference (oracle/_ref/libref_decoder.so, gradient demodulator)",
             "per_stream": [{"frames": len(f), "sha256": digest(f), "header_pos": [int(p) for p in pos],
                             "frame_sha": [hashlib.sha256(fr).hexdigest()[:10] for fr in f],
                             # (the gradient estimator is not the transmitter's inverse on every symbol - docs/LAB_NOTEBOOK.md section 2,
                             # tests/test_gpu_parity.py::test_gradient_vs_fft_divergence: the reference itself gets some payloads wrong)
                             "payloads_as_sent": sum(1 for a, b in zip([fr[15:] for fr in f], expect[k]) if a == b)} for k, (f, pos) in enumerate(res)]}
    print("%-20s %4d frames  %6.1f s  payloads as sent: %d" % (tag, sum(e["frames"] for e in entry["per_stream"]), time.time() - t0,
                                                               sum(e["payloads_as_sent"] for e in entry["per_stream"])), flush=True)
    return entry


def main():
    if not os.path.exists(os.path.join(R.REFERENCE_ROOT, "lib", "decoder_impl.cc")):
        sys.exit("needs /root/reference to build oracle/_ref")
    only = sys.argv[1:]
    path = os.path.join(HERE, "fullsize_ref.json")
    out = json.load(open(path)) if (only and os.path.exists(path)) else {}
    for c in CASES:
        if only and not any(c[0].startswith(o) for o in only):
            continue
        out[c[0]] = run_case(*c)
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
