#!/usr/bin/env python3
"""GNU-Radio-free equivalent of the reference's apps/lora_receive_file_nogui.py:
SigMF trace -> lora_receiver (channeliser + MI355X decoder) -> message_socket_sink (UDP)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gr_lora_amd import lora, sigmf  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="Decode a SigMF LoRa capture on the MI355X")
    ap.add_argument("file", nargs="?", default="example-trace", help="base name of .sigmf-data / .sigmf-meta")
    ap.add_argument("--ip", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=40868)
    ap.add_argument("--chunk", type=int, default=1 << 16, help="items per work() call")
    args = ap.parse_args()
    meta = sigmf.read_meta(args.file + ".sigmf-meta")
    cfg = sigmf.LoRaConfig(meta["transmit_freq"], meta["sf"], meta["cr"], meta["bw"], meta["prlen"], meta["crc"], meta["implicit"])
    print("[+] Configuration: %s" % cfg.string_repr())
    print("[+] Decoding. You should see a header, followed by '%s'%s %d times." % (
        meta["expected"], " and a CRC" if meta["crc"] else "", meta["times"]))
    rx = lora.lora_receiver(meta["sample_rate"], meta["capture_freq"], [meta["transmit_freq"]], cfg.bw, cfg.sf,
                            cfg.implicit, cfg.cr_num, cfg.crc)
    sink = lora.message_socket_sink(args.ip, args.port, 0)
    lora.msg_connect(rx, "frames", sink, "in")
    iq = sigmf.read_data(args.file + ".sigmf-data")
    for i in range(0, iq.size, args.chunk):
        rx.work(iq[i:i + args.chunk])
    rx.stop()
    print("[+] Done")


if __name__ == "__main__":
    main()
