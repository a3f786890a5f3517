"""N>1 path on CPU: world_size-2 gloo run of the frame gather + stream sharding."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gr_lora_amd import gather


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = gather.shard_streams(7, rank, world)
    frames = [(bytes([rank, s]) * (3 + s), s, 1000 * s + rank) for s in mine]
    allf = gather.gather_frames(frames, torch.device("cpu"))
    # the slot form bench.py exchanges (Handle.drain_slots output): same result through gather_slots
    slots, counts = gather.gather_slots(gather.pack_frames(frames, len(frames)), torch.device("cpu"))
    assert [gather.unpack_frames(slots[r], counts[r]) for r in range(world)] == allf
    q.put((rank, allf))
    dist.destroy_process_group()


def test_pack_roundtrip():
    rng = np.random.default_rng(0)
    frames = [(bytes(rng.integers(0, 256, int(rng.integers(18, 276)), dtype=np.uint8)), int(rng.integers(0, 64)), int(rng.integers(0, 1 << 40))) for _ in range(20)]
    slots = gather.pack_frames(frames, 32)
    assert gather.unpack_frames(slots, 20) == frames
    with pytest.raises(ValueError):
        gather.pack_frames(frames, 4)


def test_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        allf = res[r]
        assert len(allf) == 2
        for src in (0, 1):
            want = [(bytes([src, s]) * (3 + s), s, 1000 * s + src) for s in gather.shard_streams(7, src, 2)]
            assert allf[src] == want
    assert sorted(gather.shard_streams(7, 0, 2) + gather.shard_streams(7, 1, 2)) == list(range(7))


def test_pack_raw_equals_pack_frames():
    rng = np.random.default_rng(4)
    frames = [(bytes(rng.integers(0, 256, int(rng.integers(18, 276)), dtype=np.uint8)), int(rng.integers(0, 64)), int(rng.integers(0, 1 << 40))) for _ in range(37)]
    infos = np.zeros(len(frames), dtype=[("stream", "<u4"), ("length", "<u4"), ("header_pos", "<i8"), ("end_pos", "<i8")])
    for i, (b, s, hp) in enumerate(frames):
        infos[i] = (s, len(b), hp, 0)
    buf = np.frombuffer(b"".join(b for b, _, _ in frames), dtype=np.uint8)
    assert (gather.pack_raw(buf, infos, 40) == gather.pack_frames(frames, 40)).all()
    slots, counts = gather.gather_raw(buf, infos, torch.device("cpu"))
    assert counts == [37] and gather.unpack_frames(slots[0], 37) == frames


def test_gather_slots_single_process():
    rng = np.random.default_rng(3)
    frames = [(bytes(rng.integers(0, 256, int(rng.integers(18, 276)), dtype=np.uint8)), int(rng.integers(0, 8)), int(rng.integers(0, 1 << 40))) for _ in range(11)]
    mine = gather.pack_frames(frames, len(frames))
    slots, counts = gather.gather_slots(mine, torch.device("cpu"))
    assert counts == [11] and gather.unpack_frames(slots[0], 11) == frames


class _StubTiming:
    walker_ms = 0.25
    walker_launches = 1


class _StubHandle:
    """Stands in for gr_lora_amd.capi.Handle on a machine without a GPU: `decode` of pass k yields frames that encode
    (rank, handle, pass) so that the test can tell which pass every gathered block came from."""

    def __init__(self, rank, idx, log):
        self.rank, self.idx, self.log, self.n = rank, idx, log, 0
        self.open = False

    def decode_device_prepass(self, *a, **k):
        self.log.append(("pre", self.idx))

    def decode_device_begin(self, *a, **k):
        assert not self.open
        self.open = True
        self.log.append(("begin", self.idx))

    def decode_device_end(self):
        assert self.open
        self.open = False
        self.log.append(("end", self.idx))

    def drain_slots(self, slot_bytes):
        assert slot_bytes == gather.SLOT_BYTES
        k = self.n
        self.n += 1
        frames = [(bytes([self.rank, self.idx, k, j]) * 5, 10 * self.rank + j, 1000 * k + j) for j in range(1 + (self.rank + k) % 3)]
        return gather.pack_frames(frames, len(frames))

    def timing(self):
        return _StubTiming()


def _pipeline_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    for depth in (1, 2, 3):
        log = []
        hs = [_StubHandle(rank, i, log) for i in range(depth)]
        gat = gather.AsyncSlotGather(torch.device("cpu"), 8)
        pipe = gather.PassPipeline(hs, gat, 0, 0, [0], [0], 0)
        kept = []
        wk, ln = pipe.run(5, kept)
        assert (round(wk, 6), ln) == (1.25, 5) and len(kept) == 5
        steps = []
        for slots, counts in kept:
            assert len(counts) == world
            steps.append([gather.unpack_frames(slots[r], counts[r]) for r in range(world)])
        if depth > 1: # pipelined: pass k + 1 is begun before pass k is ended
            assert log.index(("begin", 1)) < log.index(("end", 0))
        out[depth] = steps
    q.put((rank, out))
    dist.destroy_process_group()


def test_bench_pass_pipeline_world2_gloo():
    """The bench's step loop (PassPipeline + AsyncSlotGather: one asynchronous all_gather per step, collected a step
    later) on two ranks over gloo with stub decoder handles: every step's gathered block holds exactly what each rank
    produced in THAT step, in order, for every pipeline depth."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]                                   # every rank sees the same gathered frames
    for depth in (1, 2, 3):
        for k, step in enumerate(res[0][depth]):
            for r in range(2):
                n_pass = k // depth                           # the k-th pass overall is pass k // depth of handle k % depth
                want = [(bytes([r, k % depth, n_pass, j]) * 5, 10 * r + j, 1000 * n_pass + j) for j in range(1 + (r + n_pass) % 3)]
                assert step[r] == want, (depth, k, r)


def test_async_gather_single_process():
    g = gather.AsyncSlotGather(torch.device("cpu"), 4)
    frames = [(b"abc" * 7, 3, 12345), (b"z" * 30, 1, 7)]
    g.submit(gather.pack_frames(frames, 2))
    slots, counts = g.collect()
    assert counts == [2] and gather.unpack_frames(slots[0], 2) == frames
    assert g.collect() is None
    with pytest.raises(ValueError):
        g.submit(gather.pack_frames(frames * 3, 6))


# ---- one long stream over two ranks (second clause): gather.split_stream_ranges / owned ----------------
def _split_worker(rank, world, port, q):
    import numpy as np
    from gr_lora_amd import synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.TxConfig(sf=7, cr=4)
    rng = np.random.default_rng(77)                       # (the same stream on every rank: what a shared capture is)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(4, 60)), dtype=np.uint8)) for _ in range(40)]
    st = synth.build_stream(payloads, cfg, rng=rng, gap_symbols=(0.0, 7.0))     # back-to-back packets among them: cuts fall inside packets
    for cuts in ((), [int(p) - 14 * cfg.sps for p in st.header_starts]):         # raw cuts, and cuts snapped to "gap starts"
        ranges = gather.split_stream_ranges(st.iq.size, world, cfg.sps, max_packet_symbols=8 + 8 * 20, cuts=cuts)
        start, stop, lo, hi = ranges[rank]
        o = O.Oracle(sf=7, cr=4, demod=O.DEMOD_FFT_COMPAT)  # the rank's decoder (the device decoder on a GPU; its C restatement here)
        o.run(st.iq[start:stop])
        mine = gather.owned([(f, 0, p) for f, p in zip(o.frames(), o.frame_positions())], start, lo, hi)
        allf = gather.gather_frames(mine, torch.device("cpu"))
        q.put((rank, bool(cuts), ranges, [[(b[15:], hp) for b, _s, hp in fr] for fr in allf]))
    dist.destroy_process_group()


def test_single_stream_split_over_two_ranks_equals_serial(oracle_mod):
    import numpy as np
    from gr_lora_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = synth.TxConfig(sf=7, cr=4)
    rng = np.random.default_rng(77)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(4, 60)), dtype=np.uint8)) for _ in range(40)]
    st = synth.build_stream(payloads, cfg, rng=rng, gap_symbols=(0.0, 7.0))
    o = oracle_mod.Oracle(sf=7, cr=4, demod=oracle_mod.DEMOD_FFT_COMPAT)
    o.run(st.iq)
    serial = [(f[15:], p) for f, p in zip(o.frames(), o.frame_positions())]
    assert len(serial) == 40
    for rank, snapped, ranges, per_rank in res:
        assert ranges[0][3] == ranges[1][2] and ranges[0][1] > ranges[0][3] and ranges[1][0] < ranges[1][2]   # margins on both sides of the cut
        merged = sorted(per_rank[0] + per_rank[1], key=lambda t: t[1])
        assert merged == serial, (rank, snapped, len(merged))                 # every frame once, serial bytes, serial positions
        assert len(per_rank[0]) > 5 and len(per_rank[1]) > 5
