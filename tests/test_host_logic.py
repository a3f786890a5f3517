"""CPU tests of the host side: C-ABI surface, sinks, channeliser, SigMF plumbing
(BASELINE config 1 on the oracle path -- no GPU)."""
import os
import re
import socket
import subprocess

import numpy as np
import pytest

from gr_lora_amd import lora, sigmf, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
README_BYTES = bytes.fromhex("049040deadbeef700d")


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads here (no GPU) and exports exactly what include/lora_hip.h declares."""
    from gr_lora_amd import build, capi
    build.build_library()
    hdr = open(os.path.join(ROOT, "include", "lora_hip.h")).read()
    declared = set(re.findall(r"\b(lora_hip_[a-z_]+)\s*\(", hdr))
    declared -= {"lora_hip_status"}
    assert declared == set(capi.EXPORTS)
    lib = capi.load()
    for name in declared:
        assert getattr(lib, name) is not None
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (lora_hip_[a-z_]+)", out))
    assert declared <= exported
    assert lib.lora_hip_abi_version() == 4
    # the channeliser's header
    hdr2 = open(os.path.join(ROOT, "include", "lora_hip_channelizer.h")).read()
    declared2 = set(re.findall(r"\b(lora_hip_channelizer_[a-z_]+)\s*\(", hdr2))
    assert declared2 == set(capi.EXPORTS_CHANNELIZER) and declared2 <= exported
    for name in declared2:
        assert getattr(lib, name) is not None


def test_create_fails_loudly_without_gpu_or_with_bad_sf():
    """No CPU fallback: without a device lora_hip_create returns an error (never a working handle)."""
    import ctypes as C
    from gr_lora_amd import capi
    lib = capi.load()
    cfg = capi.Config(struct_size=C.sizeof(capi.Config), samp_rate=1e6, bandwidth=125000, sf=5, cr=4, crc=1, demod=2)
    h = C.c_void_p()
    assert lib.lora_hip_create(C.byref(cfg), C.byref(h)) == -1          # BAD_SF before anything else (:57-61)
    import torch
    if not torch.cuda.is_available():
        cfg.sf = 7
        assert lib.lora_hip_create(C.byref(cfg), C.byref(h)) == -3      # NO_DEVICE
        assert not h.value
        with pytest.raises(capi.LoraHipError):
            capi.Handle(sf=7)
    with pytest.raises(SystemExit):
        lora.decoder(1e6, 125000, 13, False, 4, True)


def test_firdes_low_pass_shape():
    taps = lora.low_pass_taps(1.0, 1e6, 125000 / 2.0 + 15000.0, 10000.0)
    assert len(taps) == 241                                  # 53*fs/(22*10 kHz) -> odd
    assert abs(taps.sum() - 1.0) < 1e-5
    assert np.allclose(taps, taps[::-1])
    w = np.abs(np.fft.fft(taps, 8192))
    f = np.fft.fftfreq(8192, 1e-6)
    assert w[np.abs(f) < 60e3].min() > 0.98 and w[np.abs(f) > 100e3].max() < 0.01


def test_firdes_restatements_agree():
    """The host mirror's taps and the oracle's restatement of firdes::low_pass are the same float vector."""
    from oracle import channelizer_oracle as co
    assert np.array_equal(lora.low_pass_taps(1.0, 1e6, 125000 // 2 + 15000.0, 10000.0), co.firdes_low_pass(1.0, 1e6, 125000 // 2 + 15000.0, 10000.0))
    assert len(co.firdes_low_pass(1.0, 2e6, 77500.0, 10000.0)) == 481


def test_channelizer_oracle_streaming_equals_one_shot():
    from oracle import channelizer_oracle as co
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(50000) + 1j * rng.standard_normal(50000)).astype(np.complex64)
    a = co.Channelizer(1e6, 868.0e6, 868.1e6, 125000, 1).work(x)
    c = co.Channelizer(1e6, 868.0e6, 868.1e6, 125000, 1)
    parts, pos = [], 0
    while pos < x.size:
        n = int(rng.integers(1, 5000))
        parts.append(c.work(x[pos:pos + n]))
        pos += n
    b = np.concatenate(parts)
    assert a.size == b.size == x.size
    assert np.allclose(a, b, atol=1e-9)
    # decimation keeps its phase across chunks
    d = co.Channelizer(1e6, 868.0e6, 868.1e6, 125000, 4)
    parts = [d.work(x[i:i + 4099]) for i in range(0, x.size, 4099)]
    assert np.allclose(np.concatenate(parts), co.Channelizer(1e6, 868.0e6, 868.1e6, 125000, 4).work(x), atol=1e-9)


def test_message_socket_sink_layers():
    """lib/message_socket_sink_impl.cc:93-122: layer 0 whole blob, 1 drops loratap, 2 drops PHY header + CRC."""
    srv = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    srv.bind(("127.0.0.1", 0))
    srv.settimeout(5)
    port = srv.getsockname()[1]
    blob = bytes(15) + README_BYTES
    want = {0: blob, 1: README_BYTES, 2: bytes.fromhex("deadbeef")}
    for layer in (0, 1, 2):
        src = lora._MsgBlock()
        src.message_port_register_out("frames")
        sink = lora.message_socket_sink("127.0.0.1", port, layer)
        lora.msg_connect(src, "frames", sink, "in")
        src.message_port_pub("frames", blob)
        assert srv.recvfrom(4096)[0] == want[layer]
        sink.close()
    no_crc = bytes(15) + bytes([4, 0x80, 0x40]) + bytes.fromhex("deadbeef")
    sink = lora.message_socket_sink("127.0.0.1", port, 2)
    sink.handle(no_crc)
    assert srv.recvfrom(4096)[0] == bytes.fromhex("deadbeef")
    srv.close()


def test_config1_plumbing_on_oracle(tmp_path, oracle_mod):
    """BASELINE config 1 (CPU, no GPU): synthesised usrp-868.1-sf7-cr4-bw125-crc-0 SigMF trace at a
    100 kHz offset -> channeliser -> reference-decoder restatement -> `04 90 40 de ad be ef 70 0d` x5."""
    cfg = synth.TxConfig(sf=7, cr=4, crc=True, sync_shifts=(24, 32))
    st = synth.build_stream([bytes.fromhex("deadbeef")] * 5, cfg, rng=np.random.default_rng(868), lead=20000, tail_symbols=6)
    n = np.arange(st.iq.size, dtype=np.float64)
    rf = (st.iq * np.exp(2j * np.pi * 100e3 * n / 1e6)).astype(np.complex64)     # TX at 868.1 MHz seen from 868.0 MHz
    base = str(tmp_path / "usrp-868.1-sf7-cr4-bw125-crc-0")
    sigmf.write_trace(base, rf, 1e6, 868.0e6, 868.1e6, 7, "4/8", 125000, 8, True, False, "deadbeef", 5)
    meta = sigmf.read_meta(base + ".sigmf-meta")
    lc = sigmf.LoRaConfig(meta["transmit_freq"], meta["sf"], meta["cr"], meta["bw"], meta["prlen"], meta["crc"], meta["implicit"])
    assert lc.cr_num == 4 and lc.string_repr() == "868.1 MHz, SF 7, CR 4/8, BW 125 kHz, prlen 8, crc on, implicit off"
    from oracle import channelizer_oracle as co
    chan = co.Channelizer(meta["sample_rate"], meta["capture_freq"], meta["transmit_freq"], lc.bw, 1)
    data = sigmf.read_data(base + ".sigmf-data")
    bb = np.concatenate([chan.work(data[i:i + 65536]) for i in range(0, data.size, 65536)]).astype(np.complex64)
    frames = oracle_mod.decode_stream(bb, demod=0, sf=7, cr=lc.cr_num, crc=True)
    assert [f[15:] for f in frames] == [README_BYTES] * meta["times"]
