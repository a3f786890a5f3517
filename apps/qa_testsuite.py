#!/usr/bin/env python3
"""GNU-Radio-free counterpart of the reference's python/qa_testsuite.py (:39-254): runs every SigMF capture of a test
suite through   file -> translating low-pass (frequency offset) -> lora_receiver -> message_socket_sink(UDP, layer 2),
collects the payloads with LoRaUDPServer and writes the same text / markdown report (docs/test-results/<suite>.md).
Both filters and the decoder run on the MI355X (include/lora_hip_channelizer.h, include/lora_hip.h)."""
import argparse
import datetime
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gr_lora_amd import capi, lora, sigmf  # noqa: E402


def trunc(target, max_len=30):  # qa_testsuite.py:27-37
    if len(target) > max_len:
        return target[0:int(max_len / 2) - 1] + ".." + target[-int(max_len / 2) + 1:]
    return target


class Test:  # python/loraconfig.py
    def __init__(self, payload, times):
        self.payload, self.times = payload, times


class TestResult:
    def __init__(self, decoded_data, lora_config, test):
        self.decoded_data, self.lora_config, self.test = decoded_data, lora_config, test


class TestSummary:
    """qa_testsuite.py:39-148, same report text."""

    def __init__(self, suite):
        self.suite = suite
        now = str(datetime.datetime.utcnow())
        self._summary_text = "-------- Test suite '{:s}' results on {:s} ---------\n".format(suite, now)
        self._summary_markdown = "# Test suite: '{:s}'\n\n*Results on {:s}*\n".format(suite, now)
        self.num_total_correct_payloads = 0
        self.num_total_payloads = 0
        self._num_tests = 0
        self._last_config = None

    def add(self, test_result, print_intermediate=False):
        if not isinstance(test_result, TestResult):
            raise Exception("Test result must be of type TestResult")
        self._num_tests += 1
        lora_config, test = test_result.lora_config, test_result.test
        text, md = "", ""
        if self._last_config != vars(lora_config):
            text += "{:s}:\n".format(lora_config.string_repr())
            md += "\n### {:s}\n\nTransmitted payload | :heavy_check_mark: | :hash: | :heavy_division_sign:\n--- | --- | --- | ---\n".format(lora_config.string_repr())
            self._last_config = dict(vars(lora_config))
        num_payloads = num_correct = 0
        for i in range(test.times):
            num_payloads += 1
            self.num_total_payloads += 1
            decoded = test_result.decoded_data[i] if i < len(test_result.decoded_data) else "?"
            if isinstance(decoded, bytes):
                decoded = decoded.decode("utf-8")
            if decoded == test.payload:
                num_correct += 1
                self.num_total_correct_payloads += 1
        text += "\tTest {:>3n}: {:<30s} * {:<3n} :: passed {:>3n} out of {:<3n} ({:.2%})\n".format(
            self._num_tests, trunc(test.payload), test.times, num_correct, num_payloads, float(num_correct) / num_payloads)
        md += "`{:<30s}` | {:>3n} | {:>3n} | {:>.2%}\n".format(trunc(test.payload), num_correct, num_payloads, float(num_correct) / num_payloads)
        self._summary_text += text
        self._summary_markdown += md
        if print_intermediate:
            print(text)

    def export_summary(self, path, print_output=True, write_output=True):
        frac = float(self.num_total_correct_payloads) / max(1, self.num_total_payloads)
        self._summary_text += "\nRan a total of {:n} tests, together containing {:n} payloads.\n".format(self._num_tests, self.num_total_payloads)
        self._summary_text += "====== Total payloads passed: {:>5n} out of {:<5n}  ({:.2%}) ======\n".format(
            self.num_total_correct_payloads, self.num_total_payloads, frac)
        self._summary_markdown += "\n### Summary for suite '{:s}'\n\n".format(self.suite)
        self._summary_markdown += "Total payloads passed: {:n} out of {:n} ({:.2%})\n\n".format(self.num_total_correct_payloads, self.num_total_payloads, frac)
        if print_output:
            print(self._summary_text)
        if write_output:
            os.makedirs(path, exist_ok=True)
            with open(os.path.join(path, self.suite + ".md"), "w") as f:
                f.write(self._summary_markdown)
        return self._summary_markdown


class qa_testsuite:
    def __init__(self, path, port=40868):
        self.port = port
        self.server = lora.LoRaUDPServer(ip="127.0.0.1", port=port, timeout=3)
        self.test_suites_directory = os.path.abspath(path)
        self.reports_directory = os.path.abspath(os.path.join(path, "..", "test-results"))
        self.test_suites = sorted(x for x in os.listdir(self.test_suites_directory) if os.path.isdir(os.path.join(self.test_suites_directory, x)))

    def run(self, suites_to_run=(), write_output=True, print_output=True, chunk=1 << 18):
        summaries = {}
        for test_suite in self.test_suites:
            if suites_to_run and test_suite not in suites_to_run:
                continue
            summary = TestSummary(suite=test_suite)
            d = os.path.join(self.test_suites_directory, test_suite)
            metas = sorted((x for x in os.listdir(d) if x.endswith(".sigmf-meta")), key=lambda f: os.stat(os.path.join(d, f)).st_mtime)
            for m in metas:
                meta = sigmf.read_meta(os.path.join(d, m))
                lc = sigmf.LoRaConfig(meta["transmit_freq"], meta["sf"], meta["cr"], meta["bw"], meta["prlen"], meta["crc"], meta["implicit"])
                test = Test(meta["expected"], meta["times"])
                fs = meta["sample_rate"]
                # qa_testsuite.py:228-233: reduced rate for SF > 10, channel list [868100000], decimation 1, and a
                # freq_xlating_fir_filter(1, low_pass(1, fs, 200 kHz, 100 kHz), frequency_offset, fs) in front
                rx = lora.lora_receiver(fs, meta["capture_freq"], [868100000], lc.bw, lc.sf, False, 4, True, reduced_rate=lc.sf > 10, decimation=1)
                pre = capi.Channelizer(fs, 0.0, [float(meta["frequency_offset"])], lc.bw, 1, cutoff_hz=200000.0, transition_hz=100000.0)
                sink = lora.message_socket_sink("127.0.0.1", self.port, 2)
                lora.msg_connect(rx, "frames", sink, "in")
                iq = sigmf.read_data(os.path.join(d, m[: -len(".sigmf-meta")] + ".sigmf-data"))
                for i in range(0, iq.size, chunk):
                    rx.work(pre.work(iq[i:i + chunk])[0])
                rx.stop()
                pre.close()
                sink.close()
                decoded = self.server.get_payloads(test.times)
                summary.add(TestResult(decoded_data=decoded, lora_config=lc, test=test), print_intermediate=print_output)
            summary.export_summary(self.reports_directory, print_output=print_output, write_output=write_output)
            summaries[test_suite] = summary
        return summaries

    def close(self):
        self.server.close()


def main():
    ap = argparse.ArgumentParser(description="Tool to evaluate decoding test suites on the MI355X decoder.")
    ap.add_argument("suites", nargs="*", help="Names of the test suites to execute.")
    ap.add_argument("--path", default="./test-suites/", help="Path of the test suites")
    ap.add_argument("--nowrite", action="store_true", help="Do not write anything.")
    args = ap.parse_args()
    qa = qa_testsuite(args.path)
    qa.run(args.suites, write_output=not args.nowrite)
    qa.close()


if __name__ == "__main__":
    main()
