#!/usr/bin/env python3
"""Generates the committed golden fixtures FROM THE REFERENCE ITSELF.

oracle/_ref/libref_decoder.so is /root/reference/lib/decoder_impl.cc compiled unmodified against stand-in headers
(oracle/ref_build/); this script runs it on seeded synthetic streams (gr_lora_amd/synth.py rebuilds the same IQ from
the seeds at test time) and records what it did:

  * "ref": frames published on the `frames` port (loratap | PHY header | payload), the sample position of every
    frame's first header symbol, and the complete work() trace [state, position, consume_each, bin, d_fine_sync] -
    the reference's shipped configuration (gradient demodulator, decoder_impl.cc:499);
  * "fft": the reference's get_shift_fft (:430-464, dead code upstream, the north-star demodulator) evaluated at the
    ground-truth symbol offsets of every header / payload symbol.
  * "modes" "1"/"2": frames and bins of the FFT / FFT_COMPAT receive paths.  The reference HAS no such work() path
    (line :500 is commented out), so these come from the restated oracle and are labelled "source": "oracle".

Needs /root/reference (the build container).  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gr_lora_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import ref as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def symbol_offsets(st, cfg):
    """ground-truth first sample of every header and payload symbol"""
    out = []
    for hs, (h, q) in zip(st.header_starts, st.shifts):
        out += [hs + k * cfg.sps for k in range(len(h) + len(q))]
    return out


def make_case(sf, cr, seed, n_packets, implicit=False, nodrift=False, crc=True, lengths=(1, 24)):
    rng = np.random.default_rng(seed)
    reduced = sf > 10
    cfg = synth.TxConfig(sf=sf, cr=cr, crc=crc, reduced_rate=reduced, implicit=implicit)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(*lengths)), dtype=np.uint8)) for _ in range(n_packets)]
    gaps = [int(g) for g in rng.integers(2 * cfg.sps, 6 * cfg.sps, n_packets)]
    s = synth.build_stream(payloads, cfg, gaps=gaps)
    kw = dict(sf=sf, cr=cr, crc=crc, implicit=implicit, reduced_rate=reduced, disable_drift_correction=nodrift)
    entry = {"sf": sf, "cr": cr, "crc": crc, "implicit": implicit, "reduced_rate": reduced, "disable_drift_correction": nodrift,
             "seed": seed, "payloads": [p.hex() for p in payloads], "gaps": gaps,
             "shifts": [[h, q] for h, q in s.shifts], "n_items": int(s.iq.size)}
    r = R.Reference(**kw)
    r.enable_trace()
    r.run(s.iq)
    entry["ref"] = {"source": "reference (oracle/_ref)", "frames": [f.hex() for f in r.frames()], "header_pos": r.frame_positions(),
                    "trace": [list(t[:5]) for t in r.trace()]}
    offs = symbol_offsets(s, cfg)
    entry["fft"] = {"source": "reference get_shift_fft (oracle/_ref)", "offsets": offs,
                    "shifts": [int(r.get_shift_fft(s.iq[o:o + cfg.sps])) for o in offs]}
    entry["modes"] = {}
    for mode in (1, 2):
        o = O.Oracle(demod=mode, **kw)
        o.enable_trace()
        o.run(s.iq)
        bins = [t[3] for t in o.trace() if t[0] in (4, 5)]
        entry["modes"][str(mode)] = {"source": "oracle", "frames": [f.hex() for f in o.frames()],
                                     "header_pos": o.frame_positions(), "bins": bins}
    return entry


def main():
    if not os.path.exists(os.path.join(R.REFERENCE_ROOT, "lib", "decoder_impl.cc")):
        sys.exit("needs /root/reference to build oracle/_ref")
    cases = []
    # one small IQ file: SF7 CR4/8, the README packet twice
    cfg = synth.TxConfig(sf=7, cr=4, crc=True, sync_shifts=(24, 32))
    st = synth.build_stream([bytes.fromhex("deadbeef")] * 2, cfg, gaps=[3000, 2500], tail_symbols=2.5)
    st.iq.tofile(os.path.join(HERE, "sf7_cr4_deadbeef_x2.cf32"))
    iq_frames = [f.hex() for f in R.decode_stream(st.iq, sf=7, cr=4)]
    for sf in (7, 8, 9, 10):
        for cr in (1, 2, 3, 4):
            cases.append(make_case(sf, cr, 100 * sf + cr, 3))
    for sf in (11, 12):
        for cr in (1, 4):
            cases.append(make_case(sf, cr, 100 * sf + cr, 1, lengths=(1, 6)))
    # the two constructor switches of decoder::make that change the receive path
    for sf, cr in ((7, 4), (8, 2), (9, 3)):
        cases.append(make_case(sf, cr, 1000 + 10 * sf + cr, 2, nodrift=True))
        cases.append(make_case(sf, cr, 2000 + 10 * sf + cr, 2, implicit=True, crc=(cr != 2), lengths=(4, 20)))
    golden = {
        "generated_by": "tests/golden/make_golden.py from oracle/_ref (the reference's lib/decoder_impl.cc compiled unmodified)",
        "readme_known_answer": "049040deadbeef700d",
        "appendix_c": {"header": [29, 1, 97, 125, 37, 109, 1, 97],
                       "payload": [119, 51, 20, 1, 22, 82, 37, 58, 2, 17, 28, 115, 117, 98, 110, 7]},
        "iq_file": {"name": "sf7_cr4_deadbeef_x2.cf32", "sf": 7, "cr": 4, "gaps": [3000, 2500], "frames": iq_frames},
        "cases": cases,
    }
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(golden, f, separators=(",", ":"))
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
