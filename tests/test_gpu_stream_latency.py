"""Streaming latency of lora_hip_work (docs/LAB_NOTEBOOK.md 4.8).

The reference publishes a frame inside the work() call that completes the packet (decoder_impl.cc:870-881).  The library decodes
in device passes; lora_hip_set_stream_latency bounds how long a delivered sample may wait for its pass (wall clock), and a
finished pass is collected by the next work() call.  A 1 Msps trickle in GNU-Radio-sized calls must therefore surface every
frame within the bound + one call period, long before a 2^20-item chunk fills - and the bytes must not depend on it.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _trickle(h, iq, call_items, rate, deadline_s=30.0):
    """feeds iq at `rate` items/s in calls of call_items; returns [(blob, info, t_seen, t_last_sample_delivered)]"""
    out, pos, t0 = [], 0, time.perf_counter()
    t_delivered = []   # (end item index, wall time) per call
    while pos < iq.size:
        m = min(call_items, iq.size - pos)
        due = t0 + (pos + m) / rate                      # the call happens when its last sample exists
        while time.perf_counter() < due:
            time.sleep(0.0005)
        h.work(iq[pos:pos + m])
        pos += m
        now = time.perf_counter()
        t_delivered.append((pos, now))
        for blob, info in h.drain():
            t_last = next(t for end, t in t_delivered if end >= info.end_pos)
            out.append((blob, info, now, t_last))
        assert now - t0 < deadline_s
    return out


def test_trickle_frames_surface_within_the_bound():
    import torch
    from gr_lora_amd import capi, synth
    assert torch.cuda.is_available()
    cfg = synth.TxConfig(sf=7, cr=4)
    rng = np.random.default_rng(5)
    payloads = [bytes(rng.integers(0, 256, 24, dtype=np.uint8)) for _ in range(5)]
    st = synth.build_stream(payloads, cfg, rng=rng, gap_symbols=(6.0, 12.0), tail_symbols=40.0)
    want = [synth.expected_frame_tail(p, cfg) for p in payloads]
    assert st.iq.size < (1 << 20)                          # less than one default chunk: without a bound nothing would appear before flush

    h = capi.Handle(sf=7, cr=4)
    info0 = h.stream_info()
    assert info0.batch_items == 1 << 20 and abs(info0.max_latency_ms - 50.0) < 1e-6
    bound_ms = 40.0
    h.set_stream_latency(bound_ms)
    seen = _trickle(h, st.iq, 8192, 1.0e6)
    h.flush()
    rest = h.drain()
    si = h.stream_info()
    h.close()
    assert [b[15:] for b, _i, _t, _l in seen] + [b[15:] for b, _i in rest] == want
    assert len(seen) >= 4, "frames of a trickle must not wait for a full chunk (%d of 5 seen before flush)" % len(seen)
    assert si.passes_by_latency >= 3
    call_ms = 8192 / 1.0e6 * 1e3
    lat = [(t_seen - t_last) * 1e3 for _b, _i, t_seen, t_last in seen]
    # bound + the pass (launch in one call, collected by the next) + the two call periods around it + scheduling slack of this host
    assert max(lat) <= bound_ms + 3 * call_ms + 60.0, lat   # (observed: 45-70 ms; a full 2^20-item chunk alone would be 1000 ms)

    # bound off: same bytes, but only at flush
    h = capi.Handle(sf=7, cr=4)
    h.set_stream_latency(0.0)
    for pos in range(0, st.iq.size, 8192):
        h.work(st.iq[pos:pos + 8192])
    assert h.drain() == [] and h.stream_info().passes == 0
    h.flush()
    assert [b[15:] for b, _i in h.drain()] == want
    h.close()


def test_latency_bounded_passes_equal_batch_decode_noisy():
    """pass boundaries fall wherever the clock puts them (mid-packet included): output must equal one pass over the stream"""
    import torch
    from gr_lora_amd import capi, synth
    cfg = synth.TxConfig(sf=8, cr=2)
    rng = np.random.default_rng(17)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(5, 60)), dtype=np.uint8)) for _ in range(12)]
    st = synth.build_stream(payloads, cfg, rng=rng, gap_symbols=(1.0, 9.0), noise_sigma=synth.awgn_sigma_for_snr(40.0, cfg), tail_symbols=8.0)
    dev = torch.from_numpy(st.iq.view(np.float32)).to("cuda:0")
    hb = capi.Handle(sf=8, cr=2)
    hb.decode_device(dev.data_ptr(), st.iq.size, [0], [st.iq.size], torch.cuda.current_stream().cuda_stream)
    want = [(b, i.header_pos) for b, i in hb.drain()]
    hb.close()
    assert len(want) >= 10
    h = capi.Handle(sf=8, cr=2)
    h.set_stream_latency(2.0)                               # a pass every ~2 ms of wall clock: dozens of passes, cuts anywhere
    got = []
    for pos in range(0, st.iq.size, 5000):
        h.work(st.iq[pos:pos + 5000])
        time.sleep(0.0007)
        got += [(b, i.header_pos) for b, i in h.drain()]
    h.flush()
    got += [(b, i.header_pos) for b, i in h.drain()]
    si = h.stream_info()
    assert si.passes_by_latency >= 10 and got == want, (si.passes, si.passes_by_latency, len(got), len(want))
    # the streaming pass is lora_hip_work's own - the batch API must refuse to collect it
    h.set_stream_latency(0.0)
    h.work(st.iq[: 4 * cfg.sps])
    h.set_stream_latency(0.001)
    h.work(st.iq[4 * cfg.sps: 8 * cfg.sps])                 # launches a pass by the bound
    if h.stream_info().pass_in_flight:
        with pytest.raises(capi.LoraHipError):
            h.decode_device_end()
    h.flush()
    h.close()
