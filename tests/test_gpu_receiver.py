"""End-to-end on the GPU through the block-API mirror: SigMF trace -> lora_receiver
(channeliser + MI355X decoder) -> message_socket_sink -> UDP, scored like
python/qa_testsuite.py (hex string equality of datagrams)."""
import socket

import numpy as np
import pytest

from gr_lora_amd import synth

pytestmark = pytest.mark.gpu
README_BYTES = bytes.fromhex("049040deadbeef700d")


def test_receive_file_flow_known_answer(tmp_path, capsys):
    import torch
    assert torch.cuda.is_available()
    from gr_lora_amd import lora, sigmf
    cfg = synth.TxConfig(sf=7, cr=4, crc=True, sync_shifts=(24, 32))
    st = synth.build_stream([bytes.fromhex("deadbeef")] * 5, cfg, rng=np.random.default_rng(868), lead=20000, tail_symbols=6)
    n = np.arange(st.iq.size, dtype=np.float64)
    rf = (st.iq * np.exp(2j * np.pi * 100e3 * n / 1e6)).astype(np.complex64)
    base = str(tmp_path / "usrp-868.1-sf7-cr4-bw125-crc-0")
    sigmf.write_trace(base, rf, 1e6, 868.0e6, 868.1e6, 7, "4/8", 125000, 8, True, False, "deadbeef", 5)
    meta = sigmf.read_meta(base + ".sigmf-meta")
    srv = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    srv.bind(("127.0.0.1", 0))
    srv.settimeout(10)
    port = srv.getsockname()[1]
    rx = lora.lora_receiver(meta["sample_rate"], meta["capture_freq"], [meta["transmit_freq"]], meta["bw"], meta["sf"],
                            meta["implicit"], 4, meta["crc"])
    sink = lora.message_socket_sink("127.0.0.1", port, 2)       # layer=2 as in qa_testsuite.py:242
    lora.msg_connect(rx, "frames", sink, "in")
    frames = []
    lora.msg_connect(rx, "frames", frames.append)
    data = sigmf.read_data(base + ".sigmf-data")
    for i in range(0, data.size, 50000):
        rx.work(data[i:i + 50000])
    rx.stop()
    got = [srv.recvfrom(4096)[0].hex() for _ in range(meta["times"])]
    assert got == [meta["expected"]] * meta["times"]
    assert [f[15:] for f in frames] == [README_BYTES] * 5
    out = capsys.readouterr().out
    assert "Bins per symbol: \t128" in out and "Samples per symbol: \t1024" in out and "Decimation: \t\t8" in out
    assert out.count(" 04 90 40 de ad be ef 70 0d") == 5        # README.md:81
    srv.close()
    sink.close()


def test_decoder_block_streaming_and_warn_only_setters(capsys):
    from gr_lora_amd import lora
    cfg = synth.TxConfig(sf=9, cr=2, crc=False)
    payloads = [b"abc", b"defgh", b"ij"]
    st = synth.build_stream(payloads, cfg, rng=np.random.default_rng(3))
    dec = lora.decoder(1e6, 125000, 9, False, 4, True, False, False, verbose=False, batch_items=100000)
    assert dec.output_multiple() == 2 * 4096
    got = []
    lora.msg_connect(dec, "frames", got.append)
    dec.set_sf(10)
    dec.set_samp_rate(2e6)
    for i in range(0, st.iq.size, 33333):
        dec.work(st.iq[i:i + 33333])
    dec.stop()
    assert [g[15:] for g in got] == [synth.expected_frame_tail(p, cfg) for p in payloads]
    dec.close()
