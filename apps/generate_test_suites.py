#!/usr/bin/env python3
"""Synthetic counterpart of the reference's apps/generate_test_suites.py.

Upstream drives a LoRa module (RN2483) next to an SDR and records every (configuration, payload) cell of a suite as
a SigMF capture (generate_test_suites.py:84-138); the matrices below are its `decode_long` and `short_rn` suites
(:153-203).  The captures themselves are not part of the reference, so here the cells are SYNTHESISED with the
transmit model (gr_lora_amd/synth.py) at the same capture geometry: fs 1 MHz, centre 868.0 MHz, channel 868.1 MHz,
reduced rate for SF > 10 like the modules upstream uses (qa_testsuite.py:228-231)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gr_lora_amd import sigmf, synth  # noqa: E402

DECODE_LONG = ("decode_long", [(sf, "4/8") for sf in range(7, 13)], [("".join("%02x" % i for i in range(255)), 1)])
SHORT = ("short_rn", [(sf, "4/%d" % d) for sf in range(7, 13) for d in (8, 7, 6, 5)], [("deadbeef", 5), ("88", 1), ("ffff", 10)])


def generate(out_dir, suite, sample_rate=1e6, capture_freq=868.0e6, transmit_freq=868.1e6, frequency_offset=0, sfs=None,
             snr_db=None, seed=0):
    name, configs, tests = suite
    rng = np.random.default_rng(seed)
    made = []
    os.makedirs(os.path.join(out_dir, name), exist_ok=True)
    for sf, cr in configs:
        if sfs is not None and sf not in sfs:
            continue
        cr_num = int(cr.split("/")[1]) - 4
        cfg = synth.TxConfig(sf=sf, cr=cr_num, crc=True, reduced_rate=(sf > 10), sync_shifts=(24, 32))
        for payload_hex, times in tests:
            sigma = synth.awgn_sigma_for_snr(snr_db, cfg) if snr_db is not None else 0.0
            st = synth.build_stream([bytes.fromhex(payload_hex)] * times, cfg, rng=rng, lead=int(rng.integers(2, 9)) * cfg.sps,
                                    tail_symbols=6, noise_sigma=sigma)
            n = np.arange(st.iq.size, dtype=np.float64)
            rf = (st.iq * np.exp(2j * np.pi * (transmit_freq - capture_freq + frequency_offset) * n / sample_rate)).astype(np.complex64)
            base = os.path.join(out_dir, name, "synth-%.1f-sf%d-cr%d-bw125-crc-%d" % (transmit_freq / 1e6, sf, cr_num + 4, len(made)))
            sigmf.write_trace(base, rf, sample_rate, capture_freq, transmit_freq, sf, cr, 125000, 8, True, False, payload_hex, times,
                              frequency_offset=frequency_offset)
            made.append(base)
    return made


def main():
    ap = argparse.ArgumentParser(description="Synthesise gr-lora style test suites (SigMF)")
    ap.add_argument("-O", "--data-out", default="./test-suites/")
    ap.add_argument("-s", "--sample-rate", type=int, default=1000000)
    ap.add_argument("-f", "--frequency", type=float, default=868e6)
    ap.add_argument("-F", "--frequency-offset", type=int, default=0)
    ap.add_argument("--sf", type=int, nargs="*", default=None, help="restrict to these spreading factors")
    ap.add_argument("--snr", type=float, default=None, help="add AWGN at this in-band SNR (dB)")
    args = ap.parse_args()
    for suite in (DECODE_LONG, SHORT):
        files = generate(args.data_out, suite, args.sample_rate, args.frequency, 868.1e6, args.frequency_offset, args.sf, args.snr)
        print("[+] %s: %d captures" % (suite[0], len(files)))


if __name__ == "__main__":
    main()
