"""Frame gather over torch.distributed (RCCL over xGMI on MI355X, gloo on CPU).

Streams / packets are independent, so ranks never exchange IQ or intermediate
state: the only collective on the path is this gather of decoded frames to every
rank (rank 0 consumes it).  Fixed-size slots keep it to two all_gathers:
  slot = { u32 stream, u32 length, i64 header_pos, u8 blob[280] }   (296 bytes)
A frame blob is at most 15 + 3 + 257 = 275 bytes.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

SLOT_BLOB = 280
SLOT_BYTES = 16 + SLOT_BLOB


def pack_frames(frames: Sequence[Tuple[bytes, int, int]], capacity: int) -> np.ndarray:
    """frames: (blob, stream, header_pos) -> uint8 [capacity, SLOT_BYTES]"""
    out = np.zeros((capacity, SLOT_BYTES), dtype=np.uint8)
    for i, (blob, stream, hpos) in enumerate(frames):
        if i >= capacity:
            raise ValueError("frame slot capacity exceeded")
        if len(blob) > SLOT_BLOB:
            raise ValueError("frame blob too long")
        hdr = np.zeros(2, dtype=np.uint32)
        hdr[0], hdr[1] = stream, len(blob)
        out[i, 0:8] = hdr.view(np.uint8)
        out[i, 8:16] = np.array([hpos], dtype=np.int64).view(np.uint8)
        out[i, 16:16 + len(blob)] = np.frombuffer(blob, dtype=np.uint8)
    return out


def unpack_frames(slots: np.ndarray, count: int) -> List[Tuple[bytes, int, int]]:
    out = []
    for i in range(count):
        stream, length = slots[i, 0:8].view(np.uint32)
        hpos = int(slots[i, 8:16].view(np.int64)[0])
        out.append((bytes(slots[i, 16:16 + int(length)]), int(stream), hpos))
    return out


def gather_frames(frames: Sequence[Tuple[bytes, int, int]], device: torch.device, group=None) -> List[List[Tuple[bytes, int, int]]]:
    """All ranks contribute their frames; returns per-rank frame lists (on every rank)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [list(frames)]
    world = dist.get_world_size(group)
    cnt = torch.tensor([len(frames)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(1, max(counts))
    mine = torch.from_numpy(pack_frames(frames, cap)).to(device)
    slots = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(slots, mine, group=group)
    return [unpack_frames(s.cpu().numpy(), c) for s, c in zip(slots, counts)]


def pack_raw(buf: np.ndarray, infos: np.ndarray, capacity: int) -> np.ndarray:
    """Vectorised pack_frames for Handle.drain_raw() output."""
    n = infos.size
    if n > capacity:
        raise ValueError("frame slot capacity exceeded")
    out = np.zeros((capacity, SLOT_BYTES), dtype=np.uint8)
    if n == 0:
        return out
    lengths = infos["length"].astype(np.int64)
    if lengths.max() > SLOT_BLOB:
        raise ValueError("frame blob too long")
    out[:n, 0:4] = np.ascontiguousarray(infos["stream"]).view(np.uint8).reshape(n, 4)
    out[:n, 4:8] = np.ascontiguousarray(infos["length"]).view(np.uint8).reshape(n, 4)
    out[:n, 8:16] = np.ascontiguousarray(infos["header_pos"]).view(np.uint8).reshape(n, 8)
    starts = np.cumsum(lengths) - lengths
    row = np.repeat(np.arange(n), lengths)
    col = np.arange(int(lengths.sum())) - np.repeat(starts, lengths)
    out[row, 16 + col] = buf[: int(lengths.sum())]
    return out


def gather_raw(buf: np.ndarray, infos: np.ndarray, device: torch.device, group=None):
    """All ranks contribute their frames (raw form); returns (slots uint8 [world, cap, SLOT_BYTES], counts)."""
    if not (dist.is_available() and dist.is_initialized()):
        return pack_raw(buf, infos, max(1, infos.size))[None], [int(infos.size)]
    world = dist.get_world_size(group)
    cnt = torch.tensor([infos.size], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(1, max(counts))
    mine = torch.from_numpy(pack_raw(buf, infos, cap)).to(device)
    slots = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(slots, mine, group=group)
    return torch.stack(slots).cpu().numpy(), counts


def gather_slots(mine: np.ndarray, device: torch.device, group=None):
    """gather_raw for frames already in slot form (Handle.drain_slots(SLOT_BYTES)): no repacking on this rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return mine[None], [int(mine.shape[0])]
    world = dist.get_world_size(group)
    cnt = torch.tensor([mine.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(1, max(counts))
    padded = np.zeros((cap, SLOT_BYTES), dtype=np.uint8)
    padded[: mine.shape[0]] = mine
    t = torch.from_numpy(padded).to(device)
    slots = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(slots, t, group=group)
    return torch.stack(slots).cpu().numpy(), counts


class AsyncSlotGather:
    """The frame gather of a pipelined pass loop, off the critical path: ONE collective per step - an all_gather of a
    fixed-capacity block whose row 0 carries the count - issued asynchronously on a stream of its own and collected one
    step later; no count exchange, no `.item()`, nothing the decode stream waits for.

        g = AsyncSlotGather(device, capacity)
        g.submit(slots_k)          # uint8 [n_k, SLOT_BYTES] (Handle.drain_slots); returns at once
        ... next step's decode ...
        slots, counts = g.collect()   # step k's frames of every rank: uint8 [world, capacity, SLOT_BYTES], [n_r]

    Without torch.distributed it degenerates to handing the block back."""

    def __init__(self, device: torch.device, capacity: int, group=None):
        self.device, self.cap, self.group = device, int(capacity), group
        self.dist = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.dist else 1
        self.cuda = device.type == "cuda"
        self.stream = torch.cuda.Stream(device=device) if (self.cuda and self.dist) else None
        pin = self.cuda
        self.h_in = [torch.zeros((self.cap + 1, SLOT_BYTES), dtype=torch.uint8, pin_memory=pin) for _ in range(2)]
        self.h_out = [torch.zeros((self.world, self.cap + 1, SLOT_BYTES), dtype=torch.uint8, pin_memory=pin) for _ in range(2)]
        if self.dist:
            self.d_in = [torch.zeros((self.cap + 1, SLOT_BYTES), dtype=torch.uint8, device=device) for _ in range(2)]
            self.d_out = [torch.zeros((self.world, self.cap + 1, SLOT_BYTES), dtype=torch.uint8, device=device) for _ in range(2)]
        self.k = 0
        self.pending = None   # (buffer index, work handle, copy-done event)

    def submit(self, mine: np.ndarray):
        if self.pending is not None:
            raise RuntimeError("collect() the previous step first")
        n = int(mine.shape[0])
        if n > self.cap:
            raise ValueError("frame slot capacity exceeded (%d > %d)" % (n, self.cap))
        b = self.k & 1
        self.k += 1
        hin = self.h_in[b].numpy()
        hin[0, 0:8] = np.array([n], dtype=np.int64).view(np.uint8)
        if n:
            hin[1:1 + n] = mine
        if not self.dist:
            self.pending = (b, None, None)
            return
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self.d_in[b].copy_(self.h_in[b], non_blocking=True)
                work = dist.all_gather_into_tensor(self.d_out[b].view(-1), self.d_in[b].view(-1), group=self.group, async_op=True)
                work.wait()                                  # (orders the copy below behind the collective on this stream; the host does not block)
                self.h_out[b].copy_(self.d_out[b], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self.pending = (b, None, ev)
        else:                                                # CPU / gloo
            self.d_in[b].copy_(self.h_in[b])
            work = dist.all_gather_into_tensor(self.d_out[b].view(-1), self.d_in[b].view(-1), group=self.group, async_op=True)
            self.pending = (b, work, None)

    def collect(self):
        if self.pending is None:
            return None
        b, work, ev = self.pending
        self.pending = None
        if not self.dist:
            blk = self.h_in[b].numpy()[None]
        else:
            if work is not None:
                work.wait()
                self.h_out[b].copy_(self.d_out[b])
            if ev is not None:
                ev.synchronize()
            blk = self.h_out[b].numpy()
        counts = [int(blk[r, 0, 0:8].view(np.int64)[0]) for r in range(blk.shape[0])]
        return blk[:, 1:].copy(), counts   # (a copy: the internal double buffers are reused two steps on)


class PassPipeline:
    """The software-pipelined pass loop bench.py times (and a streaming receiver runs): `depth` decoder handles alternate
    on one HIP stream - begin(k+1) (plan + launch) is issued before end(k) (wait, stitch, frames), with depth 3 the
    envelope pre-pass of k+2 ahead of that - and the frames of step k go into an AsyncSlotGather that is collected
    while step k+1 runs.  `handles` need decode_device_begin / _prepass / _end, drain_slots and timing (gr_lora_amd.capi.Handle;
    the CPU tests use a stub)."""

    def __init__(self, handles, gatherer: AsyncSlotGather, dev_ptr: int, n_items: int, offs, lens, stream=0):
        """`stream`: one HIP stream handle, or a list of them - pass k then goes to stream k mod len(list): with two streams the
        walker kernel of pass k+1 starts on the CUs pass k's shorter jobs have already left (bench.py --overlap)."""
        self.hs, self.gat, self.depth = list(handles), gatherer, len(handles)
        self.streams = list(stream) if isinstance(stream, (list, tuple)) else [stream]
        self.args = (dev_ptr, n_items, offs, lens)

    def _begin(self, k):   # the IQ is resident and unchanged: the envelope pre-pass need not wait for the stream (IQ_READY)
        self.hs[k % self.depth].decode_device_begin(*self.args, self.streams[k % len(self.streams)], iq_ready=True)

    def _prepass(self, k):
        self.hs[k % self.depth].decode_device_prepass(*self.args, self.streams[k % len(self.streams)], iq_ready=True)

    def _finish(self, k):
        hk = self.hs[k % self.depth]
        hk.decode_device_end()
        done = self.gat.collect()                             # step k-1's frames of every rank (None on the first step)
        self.gat.submit(hk.drain_slots(SLOT_BYTES))           # step k's frames: one asynchronous all_gather
        return done, hk.timing()

    def run(self, n_steps: int, keep=None, check=None):
        """n_steps passes; returns (sum of walker kernel ms, walker launches).  `keep` (a list) receives every step's
        gathered (slots, counts); `check` (a callable) is handed every step's gathered block as it is collected."""
        wk, ln = 0.0, 0
        if n_steps <= 0:
            return wk, ln
        depth = self.depth
        self._begin(0)
        if depth > 2 and n_steps > 1:
            self._prepass(1)
        for k in range(n_steps):
            if depth > 2 and k + 2 < n_steps:
                self._prepass(k + 2)
            if depth > 1 and k + 1 < n_steps:
                self._begin(k + 1)
            done, tm = self._finish(k)
            if keep is not None and done is not None:
                keep.append(done)
            if check is not None and done is not None:
                check(done)
            wk += tm.walker_ms
            ln += tm.walker_launches
            if depth == 1 and k + 1 < n_steps:
                self._begin(k + 1)
        last = self.gat.collect()                             # the final step's gather belongs to the run too
        if keep is not None and last is not None:
            keep.append(last)
        if check is not None and last is not None:
            check(last)
        return wk, ln


def shard_streams(n_streams: int, rank: int, world: int) -> List[int]:
    """Static partition: stream c -> rank c mod world."""
    return [c for c in range(n_streams) if c % world == rank]


# ---- one long stream over several ranks (second clause) ---------------------------------------------------
def split_stream_ranges(n_items: int, world: int, sps: int, max_packet_symbols: int, guard_symbols: int = 16, cuts: Sequence[int] = ()):
    """Splits ONE stream of n_items over `world` ranks: rank r decodes items [start_r, stop_r) with a decoder of its own
    and KEEPS the frames whose first header symbol lies in [own_lo_r, own_hi_r).  Returns [(start, stop, own_lo, own_hi)].

    own_hi_r = own_lo_{r+1} = the cut between the ranks: n_items * (r + 1) / world, moved forward to the next of `cuts`
    (gap starts from the envelope pre-pass, Handle.gap_starts_device) when one lies within a quarter of a share - a cut
    inside a gap meets the next rank's decoder where the serial decoder is too: idle in DETECT.  A rank starts guard_symbols
    (>= the 12.25 symbols of preamble + SFD) before its first owned header position, so that it sees the whole preamble of
    every packet it owns, and runs max_packet_symbols past its last one, so that it sees the end of it; frames decoded in
    those margins belong to the neighbour and are dropped (`owned`).  Unlike the segment stitcher inside one GPU
    (lora_stitch.hpp) nothing is probed across ranks: a rank's decoder starts fresh, so the loratap SNR byte (from the last
    four DETECT windows, decoder_impl.cc:360,:377-383) of its first frames and a header decoded with a stale d_phdr.cr (:655)
    can differ from the serial decoder's; header positions and frame bytes behind the loratap header are the serial ones WHEN THE CUT LIES
    IN A GAP (pass `cuts`; bench.py --split does).  A cut inside a packet train has no such guarantee: the fresh decoder may trigger on the
    neighbour's payload and acquire the first owned packet through another SYNC window."""
    if world <= 0 or n_items < 0:
        raise ValueError("bad split")
    cuts = sorted(int(c) for c in cuts)
    share = n_items / world
    bounds = [0]
    for r in range(1, world):
        b = int(round(r * share))
        nxt = [c for c in cuts if b <= c <= b + share / 4]
        b = nxt[0] if nxt else b
        bounds.append(max(bounds[-1], min(b, n_items)))
    bounds.append(n_items)
    out = []
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        out.append((max(0, lo - guard_symbols * sps) if r else 0, min(n_items, hi + max_packet_symbols * sps) if r + 1 < world else n_items, lo, hi))
    return out


def owned(frames: Sequence[Tuple[bytes, int, int]], start: int, own_lo: int, own_hi: int) -> List[Tuple[bytes, int, int]]:
    """frames = (blob, stream, header_pos relative to the rank's range start) -> the rank's own ones with ABSOLUTE header_pos"""
    return [(b, s, hp + start) for b, s, hp in frames if own_lo <= hp + start < own_hi]
