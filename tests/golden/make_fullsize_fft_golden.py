#!/usr/bin/env python3
"""Full-size FFT-mode fixtures for the config-3 cells the GPU suite cannot afford to run the oracle on.

tests/test_gpu_fullsize.py::test_config3_full_size runs the CPU oracle beside the GPU for every cell of BASELINE config 3 in the
FFT demodulator - except SF10 / SF11 / SF12 at CR 4/6 and 4/7, where the oracle's O(sps^2) SYNC (decoder_impl.cc:392-413 restated as it
stands) makes a 256-packet cell minutes of host time.  Those six cells are pre-computed HERE, offline, by the parity build of the
oracle (oracle/liblora_oracle.so: -O2, no contraction - the build tests/test_ref_pin.py holds bit for bit to the compiled reference),
demodulator DEMOD_FFT_COMPAT, one decoder per stream; bench.make_workload rebuilds the IQ from the seeds at test time.  Recorded per
stream: frame count, sha256 over the published frames (15 B loratap | 3 B PHY header | payload) and every frame's header position.

Needs nothing outside the repo.  Run from the repo root:  python tests/golden/make_fullsize_fft_golden.py [tag-prefix ...]
"""
import concurrent.futures as cf
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "fullsize_oracle_fft.json")

# (tag, sf, cr, packets, payload, streams, seed): the calls tests/test_gpu_fullsize.py makes for config 3
CASES = [("config3-sf%d-cr%d" % (sf, cr), sf, cr, 256, 32, 8, 100 * sf + cr) for sf in (10, 11, 12) for cr in (2, 3)]


def digest(frames):
    h = hashlib.sha256()
    for f in frames:
        h.update(len(f).to_bytes(4, "little"))
        h.update(f)
    return h.hexdigest()


def run_case(tag, sf, cr, packets, payload, streams, seed):
    cfg, iq, offs, lens, expect = bench.make_workload(sf, cr, packets, payload, streams, seed=seed)
    kw = dict(sf=sf, cr=4, crc=True, reduced_rate=(sf > 10))

    def one(k):
        o = O.Oracle(demod=O.DEMOD_FFT_COMPAT, **kw)
        o.run(iq[offs[k]:offs[k] + lens[k]])
        return o.frames(), o.frame_positions()

    t0 = time.time()
    with cf.ThreadPoolExecutor(min(8, len(offs))) as ex:      # (the ctypes calls release the GIL)
        res = list(ex.map(one, range(len(offs))))
    entry = {"tag": tag, "sf": sf, "cr": cr, "packets": packets, "payload": payload, "streams": streams, "seed": seed, "decoder_kw": kw, "demod": 2,
             "n_items": int(iq.size), "source": "oracle/liblora_oracle.so (parity build), DEMOD_FFT_COMPAT",
             "per_stream": [{"frames": len(f), "sha256": digest(f), "header_pos": [int(p) for p in pos],
                             "payloads_as_sent": sum(1 for a, b in zip([fr[15:] for fr in f], expect[k]) if a == b)} for k, (f, pos) in enumerate(res)]}
    print("%-20s %4d frames  %6.1f s  payloads as sent: %d" % (tag, sum(e["frames"] for e in entry["per_stream"]), time.time() - t0,
                                                               sum(e["payloads_as_sent"] for e in entry["per_stream"])), flush=True)
    return entry


def main():
    only = sys.argv[1:]
    out = json.load(open(PATH)) if os.path.exists(PATH) else {}
    for c in CASES:
        if only and not any(c[0].startswith(o) for o in only):
            continue
        out[c[0]] = run_case(*c)
        json.dump(out, open(PATH, "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
