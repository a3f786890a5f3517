#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s demodulated on MI355X (BASELINE.json metric).

One "step" = one full pass of the decoder hot path (detect -> sync -> SFD ->
dechirp x FFT x argmax per symbol -> fine sync -> gray / deinterleave / dewhiten /
Hamming -> frames) over one batch of synthetic IQ that is already resident in HBM.
Workload at N=1: BASELINE.json configs[1] -- SF7, CR4/8, BW125k, fs 1 MHz,
1024 synthetic packets x 32-byte payload.  With N>1 every rank decodes its own
batch of the same shape (weak scaling; packets/streams are independent, the only
collective is the RCCL gather of decoded frames).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def make_workload(sf, cr, n_packets, payload_len, n_streams, seed):
    from gr_lora_amd import synth
    cfg = synth.TxConfig(sf=sf, cr=cr, crc=True, reduced_rate=(sf > 10))
    rng = np.random.default_rng(seed)
    per = n_packets // n_streams
    pieces, offs, lens, expect = [], [], [], []
    off = 0
    for s in range(n_streams):
        payloads = [bytes(rng.integers(0, 256, payload_len, dtype=np.uint8)) for _ in range(per)]
        st = synth.build_stream(payloads, cfg, rng=rng, gap_symbols=(2.0, 6.0))
        pieces.append(st.iq)
        offs.append(off)
        lens.append(st.iq.size)
        off += st.iq.size
        expect.append([synth.expected_frame_tail(p, cfg) for p in payloads])
    return cfg, np.concatenate(pieces), offs, lens, expect


def cpu_baseline(cfg, iq, offs, lens, budget_s=15.0):
    """Times the CPU oracle (restatement of the reference decoder, default gradient
    demodulator) on a bounded sample of the same workload; 1 thread like the
    reference's single GNU Radio block thread."""
    from oracle import oracle as O
    O.build()
    out = {}
    for name, mode in (("grad", O.DEMOD_GRAD), ("fft", O.DEMOD_FFT_COMPAT)):
        done = 0
        t_used = 0.0
        frames = 0
        for o_, l_ in zip(offs, lens):
            dec = O.Oracle(samp_rate=cfg.samp_rate, bandwidth=cfg.bw, sf=cfg.sf, cr=4, crc=True,
                           reduced_rate=cfg.reduced_rate, demod=mode)
            seg = iq[o_:o_ + l_]
            cap = min(l_, 40_000_000)
            t0 = time.perf_counter()
            dec.run(seg[:cap])
            t_used += time.perf_counter() - t0
            done += cap
            frames += len(dec.frames())
            if t_used > budget_s / 2:
                break
        out[name] = (done / t_used / 1e6, done, frames)
    # every host core at once: one decoder instance per stream, as a GNU Radio flowgraph with one block thread per
    # channel would run (the ctypes calls release the GIL), on a bounded slice of each stream
    import concurrent.futures as cf
    ncores = os.cpu_count() or 1
    nthreads = max(1, min(ncores, len(offs)))
    cap = int(min(min(lens), 12_000_000))

    def one(k):
        dec = O.Oracle(samp_rate=cfg.samp_rate, bandwidth=cfg.bw, sf=cfg.sf, cr=4, crc=True, reduced_rate=cfg.reduced_rate, demod=O.DEMOD_GRAD)
        dec.run(iq[offs[k]:offs[k] + cap])
        return len(dec.frames())

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(nthreads) as ex:
        list(ex.map(one, range(nthreads)))
    dt = time.perf_counter() - t0
    out["all_cores"] = (nthreads * cap / dt / 1e6, nthreads, ncores)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--sf", type=int, default=7)
    ap.add_argument("--cr", type=int, default=4)
    ap.add_argument("--packets", type=int, default=1024)
    ap.add_argument("--payload", type=int, default=32)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("LORA_BENCH_STREAMS", "8")))
    ap.add_argument("--demod", type=int, default=2, help="0 grad, 1 fft, 2 fft_compat")
    ap.add_argument("--depth", type=int, default=3, help="pipeline depth: 1 = strictly one pass after the other; 2 = while the device runs "
                    "step k+1 the host stitches step k (decoder handles alternating on one stream; walker kernels never overlap); "
                    "3 = also the envelope pre-pass of step k+2 is issued ahead (it runs in the tail of step k's walker)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gr_lora_amd import capi, gather

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg, iq, offs, lens, expect = make_workload(args.sf, args.cr, args.packets, args.payload, args.streams, seed=2 + 1000 * rank)
    n_items = int(iq.size)
    d_iq = torch.from_numpy(iq.view(np.float32)).to(dev)
    depth = max(1, min(3, args.depth))
    hs = [capi.Handle(samp_rate=cfg.samp_rate, bandwidth=cfg.bw, sf=cfg.sf, cr=4, crc=True, reduced_rate=cfg.reduced_rate,
                      device=local_rank, demod=args.demod) for _ in range(depth)]
    h = hs[0]
    stream = torch.cuda.current_stream().cuda_stream
    gather_stream = torch.cuda.Stream(device=dev) if world > 1 else None

    # One step = one full pass over the batch.  With depth 2 the passes are software-pipelined the way a streaming receiver
    # runs them: the plan + launch of step k+1 (begin) is issued before the results of step k are collected (finish), on
    # the same HIP stream, so the device goes from one walker kernel straight to the next while the host stitches.
    def begin(k):   # the IQ is resident and unchanged: the envelope pre-pass need not wait for the stream (IQ_READY)
        hs[k % depth].decode_device_begin(d_iq.data_ptr(), n_items, offs, lens, stream, iq_ready=True)

    def prepass(k):
        hs[k % depth].decode_device_prepass(d_iq.data_ptr(), n_items, offs, lens, stream, iq_ready=True)

    def finish(k):
        hk = hs[k % depth]
        hk.decode_device_end()
        mine = hk.drain_slots(gather.SLOT_BYTES)             # frames straight into the exchange layout
        if gather_stream is not None:                        # RCCL all_gather of the frames when N > 1, on a stream of its own:
            with torch.cuda.stream(gather_stream):           # on the decode stream it would queue behind the next step's kernel
                slots, counts = gather.gather_slots(mine, dev)
        else:
            slots, counts = gather.gather_slots(mine, dev)
        return slots, counts, hk.timing()

    def run(n_steps):
        wk, ln = 0.0, 0
        if n_steps <= 0:
            return wk, ln
        begin(0)
        if depth > 2 and n_steps > 1:
            prepass(1)
        for k in range(n_steps):
            if depth > 2 and k + 2 < n_steps:
                prepass(k + 2)
            if depth > 1 and k + 1 < n_steps:
                begin(k + 1)
            _s, _c, tm = finish(k)
            wk += tm.walker_ms
            ln += tm.walker_launches
            if depth == 1 and k + 1 < n_steps:
                begin(k + 1)
        return wk, ln

    # correctness of what is being timed (outside the timed region): frames as gathered, this rank's share, every handle
    verified = True
    for j in range(depth):
        begin(j)
        slots, counts, _tm = finish(j)
        mine = gather.unpack_frames(slots[rank if len(counts) > 1 else 0], counts[rank if len(counts) > 1 else 0])
        got = {}
        for b, sid, _hp in mine:
            got.setdefault(sid, []).append(b[15:])
        verified = verified and all(got.get(s, []) == expect[s] for s in range(len(offs)))

    run(40)            # pre-roll, untimed like the check above: ~20 ms of passes bring the device to its sustained clocks
    run(args.warmup)   # the W warm-up steps proper
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    walker_ms, launches = run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_items], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_items = int(tot.item())
        v = torch.tensor([1 if verified else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        verified = bool(v.item())
    else:
        total_items = n_items

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_items * args.steps / elapsed / 1e6
        kernel_ms = walker_ms / max(1, args.steps)  # walker kernel time per pass (HIP events, launch stream)
        achieved = 8.0 * n_items / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # HBM bytes per pass from the committed rocprofv3 PMC run of this same workload (separate FETCH_SIZE /
        # WRITE_SIZE passes, gfx950 FETCH x2 correction as MI355X_MICROARCH.md prescribes); null otherwise
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_f_pmc_traffic.json")))
            if pmc.get("workload_items") == n_items and args.demod != 0:
                traffic = int(pmc["hbm_bytes_per_pass_corrected"])
        except (OSError, ValueError, KeyError):
            pass
        fast = args.sf in (7, 8) and args.demod != 0 and not os.environ.get("LORA_HIP_NO_FAST")
        kname = ("walker2_kernel_sf%d" % args.sf) if fast else "walker_kernel"
        res = {
            "metric": "IQ Msamples/s demodulated", "value": round(value, 3), "unit": "Msamples/s",
            "symbols_per_s": round(value * 1e6 / cfg.sps, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SF%d CR4/%d BW125k fs1M, %d packets x %d B payload per GPU, %d stream(s)" %
                                   (args.sf, 4 + args.cr, args.packets, args.payload, args.streams),
                       "items_per_gpu": n_items, "demod": ["grad", "fft", "fft_compat"][args.demod],
                       "bit_exact_vs_expected": verified, "parallelism": "streams sharded, dp%d" % world,
                       "pipeline_depth": depth},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_unit": "HBM bytes per pass (PMC, profiles/r01_f_pmc_traffic.json)",
                         "kernel": kname, "kernel_ms_per_pass": round(kernel_ms, 4),
                         "launches_per_pass": launches / max(1, args.steps),
                         "algorithmic_bytes_per_pass": 8 * n_items},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(cfg, iq, offs, lens)
            res["cpu_baseline"] = {"value": round(cb["grad"][0], 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
                                   "sample": "oracle (C restatement, default gradient demod) over the first %d items of the same workload; "
                                             "fft demod: %.3f Msamples/s" % (cb["grad"][1], cb["fft"][0]),
                                   "all_cores": {"value": round(cb["all_cores"][0], 3), "unit": "Msamples/s", "threads": cb["all_cores"][1],
                                                 "host_cores": cb["all_cores"][2], "sample": "one decoder per stream, gradient demod, 12e6 items each"}}
        print(json.dumps(res))
    for hk in hs:
        hk.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
