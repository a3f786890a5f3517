#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s demodulated on MI355X (BASELINE.json metric).

One "step" = one full pass of the decoder hot path (detect -> sync -> SFD -> dechirp x FFT x argmax per symbol -> fine
sync -> gray / deinterleave / dewhiten / Hamming -> frames -> frame gather) over one batch of synthetic IQ.

Workloads (--config):
  2 (default, the headline)  BASELINE.json configs[1]: SF7 CR4/8 BW125k fs 1 MHz, 1024 packets x 32-byte payload
  3                          one cell of the SF sweep: --sf S, 256 packets x 32 B (reduced rate for SF > 10)
  4                          64-channel SF9 gateway: 8 continuous back-to-back streams PER GPU (channel c -> rank c mod N,
                             gr_lora_amd.gather.shard_streams), random 16-64 B payloads, seed = channel id, >= 2 s per stream
Paths (--path):
  device (default)  the IQ is resident in HBM when the timed region starts (`value` as the contract defines it)
  work              the reference block's own contract: host buffers through lora_hip_work() - PCIe included; reported as
                    its own metric, never as the headline

`--gpus N` with no launcher around it re-executes itself under torch.distributed.run (one rank per GPU, RCCL); under a
launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  Weak scaling: every rank decodes a batch of the same shape; the only
collective is the frame gather (one all_gather per step, asynchronous, collected one step later).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import re
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


# ------------------------------------------------------------------------------------------------ workloads
def _cached(key, build):
    """LORA_BENCH_CACHE=<dir>: keep a synthesised workload (IQ + expected frames) between runs on one box - same-box A/B of
    library variants (tools/ab.sh) then pays for the synthesis once"""
    d = os.environ.get("LORA_BENCH_CACHE")
    if not d:
        return build()
    import pickle
    f_iq, f_meta = os.path.join(d, key + ".npy"), os.path.join(d, key + ".pkl")
    if os.path.exists(f_iq) and os.path.exists(f_meta):
        with open(f_meta, "rb") as f:
            return (np.load(f_iq),) + pickle.load(f)
    iq, offs, lens, expect = build()
    os.makedirs(d, exist_ok=True)
    np.save(f_iq, iq)
    with open(f_meta, "wb") as f:
        pickle.dump((offs, lens, expect), f)
    return iq, offs, lens, expect


def make_workload(sf, cr, n_packets, payload_len, n_streams, seed, samp_rate=1e6):
    """config 2 / 3: n_packets packets of payload_len bytes in n_streams streams, zero gaps of 2-6 symbols"""
    from gr_lora_amd import synth
    cfg = synth.TxConfig(sf=sf, cr=cr, crc=True, reduced_rate=(sf > 10), samp_rate=samp_rate)
    key = "wl-sf%d-cr%d-%dx%dB-%dstreams-seed%d" % (sf, cr, n_packets, payload_len, n_streams, seed) + ("" if samp_rate == 1e6 else "-fs%d" % int(samp_rate)) + ("-noise%s" % os.environ["LORA_BENCH_NOISE_DB"] if "LORA_BENCH_NOISE_DB" in os.environ else "")
    return (cfg,) + tuple(_cached(key, lambda: _make_workload(cfg, n_packets, payload_len, n_streams, seed)))


def _make_workload(cfg, n_packets, payload_len, n_streams, seed):
    from gr_lora_amd import synth
    rng = np.random.default_rng(seed)
    per = n_packets // n_streams
    pieces, offs, lens, expect = [], [], [], []
    off = 0
    for s in range(n_streams):
        payloads = [bytes(rng.integers(0, 256, payload_len, dtype=np.uint8)) for _ in range(per)]
        # LORA_BENCH_NOISE_DB=<in-band SNR> (diagnostics; BASELINE's workloads are noiseless): AWGN over the whole stream, the idle gaps included.  It shows the limit of
        # the speculative segments: with ANY noise the reference's header position depends on its DETECT alignment by +-1 sample, the stitch
        # accepts a speculative job only at the true trajectory's header sample, and a mismatch is decoded again serially
        st = synth.build_stream(payloads, cfg, rng=rng, gap_symbols=(2.0, 6.0), noise_sigma=(synth.awgn_sigma_for_snr(float(os.environ["LORA_BENCH_NOISE_DB"]), cfg) if "LORA_BENCH_NOISE_DB" in os.environ else 0.0))
        pieces.append(st.iq)
        offs.append(off)
        lens.append(st.iq.size)
        off += st.iq.size
        expect.append([synth.expected_frame_tail(p, cfg) for p in payloads])
    return np.concatenate(pieces), offs, lens, expect


def make_gateway_workload(channels, seconds=2.0, sf=9):
    """config 4: every channel is an independent continuous stream (own decoder state): back-to-back packets, random
    16-64 byte payloads, seed = channel id, at least `seconds` of signal at fs = 1 MHz"""
    from gr_lora_amd import synth
    cfg = synth.TxConfig(sf=sf, cr=4, crc=True)
    want_items = int(seconds * cfg.samp_rate)
    pieces, offs, lens, expect = [], [], [], []
    off = 0
    for ch in channels:
        rng = np.random.default_rng(ch)
        payloads, n = [], 0
        while n < want_items:
            p = bytes(rng.integers(0, 256, int(rng.integers(16, 65)), dtype=np.uint8))
            payloads.append(p)
            n += (12.25 + 8 + synth.payload_symbol_count(len(p) + 2, sf, 4, False)) * cfg.sps
        st = synth.build_stream(payloads, cfg, gaps=[2 * cfg.sps] + [0] * (len(payloads) - 1), tail_symbols=2.5, rng=np.random.default_rng(7000 + ch),
                                noise_sigma=(synth.awgn_sigma_for_snr(float(os.environ["LORA_BENCH_NOISE_DB"]), cfg) if "LORA_BENCH_NOISE_DB" in os.environ else 0.0))   # (LORA_BENCH_NOISE_DB: see _make_workload)
        pieces.append(st.iq)
        offs.append(off)
        lens.append(st.iq.size)
        off += st.iq.size
        expect.append([synth.expected_frame_tail(p, cfg) for p in payloads])
    return cfg, np.concatenate(pieces), offs, lens, expect


# --------------------------------------------------------------------------------------------- CPU baseline
def synth_payload_symbols(n_bytes, cfg):
    from gr_lora_amd import synth
    return synth.payload_symbol_count(n_bytes, cfg.sf, cfg.cr, cfg.reduced_rate)


def cpu_baseline(cfg, iq, offs, lens, budget_s=20.0):
    """The reference decoder's CPU path timed beside the GPU: the C restatement of lib/decoder_impl.cc (oracle/), built
    -O3 -march=native on THIS host, same input already in RAM, steady clock around the stream -> frames
    call, median of 5 runs on a bounded sample; (i) one thread per stream (the reference decoder is one GNU Radio block
    thread), both demodulators, (ii) every host core at once, one decoder per stream."""
    from oracle import oracle as O
    O.lib_fast()
    out = {}
    sample = min(int(lens[0]), 24_000_000)
    seg = np.ascontiguousarray(iq[offs[0]:offs[0] + sample])
    for name, mode in (("grad", O.DEMOD_GRAD), ("fft", O.DEMOD_FFT_COMPAT)):
        ts, frames = [], 0
        t_begin = time.perf_counter()
        for _ in range(5):
            dec = O.Oracle(samp_rate=cfg.samp_rate, bandwidth=cfg.bw, sf=cfg.sf, cr=4, crc=True, reduced_rate=cfg.reduced_rate, demod=mode, fast=True)
            t0 = time.perf_counter()
            dec.run(seg)
            ts.append(time.perf_counter() - t0)
            frames = len(dec.frames())
            if time.perf_counter() - t_begin > budget_s / 3 and len(ts) >= 3:
                break
        out[name] = (sample / float(np.median(ts)) / 1e6, sample, frames, len(ts))
    # the reference's own lib/decoder_impl.cc (oracle/_ref, prebuilt where /root/reference exists: -O3 -march=x86-64-v3, VOLK
    # and liquid-dsp replaced by stand-ins of plain loops) over the same sample, when the library travelled with the tree
    out["reference_build"] = None
    try:
        from oracle import ref as R
        if os.path.exists(R._LIB_FAST_PATH) or R.available():
            ts = []
            for _ in range(3):
                dec = R.Reference(samp_rate=cfg.samp_rate, bandwidth=cfg.bw, sf=cfg.sf, cr=4, crc=True, reduced_rate=cfg.reduced_rate, fast=True)
                t0 = time.perf_counter()
                dec.run(seg)
                ts.append(time.perf_counter() - t0)
            out["reference_build"] = (sample / float(np.median(ts)) / 1e6, len(ts))
    except Exception as e:   # the checker library is optional here
        out["reference_build_error"] = repr(e)
    import concurrent.futures as cf
    ncores = os.cpu_count() or 1
    nthreads = max(1, min(ncores, len(offs)))
    cap = int(min(min(lens), 12_000_000))

    def one(k):
        dec = O.Oracle(samp_rate=cfg.samp_rate, bandwidth=cfg.bw, sf=cfg.sf, cr=4, crc=True, reduced_rate=cfg.reduced_rate, demod=O.DEMOD_GRAD, fast=True)
        dec.run(iq[offs[k]:offs[k] + cap])
        return len(dec.frames())

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(nthreads) as ex:   # (the ctypes calls release the GIL)
        list(ex.map(one, range(nthreads)))
    dt = time.perf_counter() - t0
    out["all_cores"] = (nthreads * cap / dt / 1e6, nthreads, ncores)
    return out


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\s+", "", text)


# what a decode pass is made of: the kernels, the device structs, the scheduler and the runtime that plans and launches
HASHED_SOURCES = ("lora_device.h", "lora_kernels.hip", "lora_runtime.cpp", "lora_stitch.hpp", "lora_walker2.inc.hip",
                  "lora_walker3.inc.hip", "lora_team_demod.inc.hip", "lora_wave_demod.inc.hip", "lora_wave_decim.inc.hip", "lora_detect.inc.hip", "lora_strict_sync.inc.hip", "lora_strict_resolve_lds.inc", "whitening_data.inc")


def source_hash(raw=False):
    """sha256 over the sources of the decode pass: ties a quoted PMC figure to the code it was measured on (the GPU box
    has no .git to ask).  Comments and white space do not count (raw=True: the files byte for byte)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gr_lora_amd", "csrc")
    for name in HASHED_SOURCES:
        with open(os.path.join(d, name), "rb") as f:
            data = f.read()
        h.update(name.encode() + b"\0" + (data if raw else _strip_comments(data.decode("utf-8", "replace")).encode()))
    return h.hexdigest()[:16]


def quoted_traffic(workload_key):
    """HBM bytes per pass from a committed rocprofv3 PMC run of this workload (profiles/*pmc_traffic*.json; separate
    FETCH_SIZE / WRITE_SIZE passes, gfx950 FETCH x2 correction as MI355X_MICROARCH.md prescribes) - only when that run
    was taken on the sources this process is running; null otherwise"""
    best = None
    pd = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pd)) if os.path.isdir(pd) else []:
        if "pmc_traffic" not in name or not name.endswith(".json"):
            continue
        try:
            pmc = json.load(open(os.path.join(pd, name)))
        except (OSError, ValueError):
            continue
        if pmc.get("workload_key") == workload_key and pmc.get("source_hash") == source_hash():
            best = (int(pmc["hbm_bytes_per_pass_corrected"]), name, pmc.get("rocprof_walker_avg_ms_per_pass"), pmc.get("rocprof_kernel_stats"))
    return best


def quoted_ceiling(sf, demod):
    """The standalone symbol demodulator's streaming rate for this SF (profiles/*demod_ceiling*.json, tools/demod_ceiling.py under
    rocprofv3: every window independent, no state machine, no acquisition) as a fraction of HBM peak - what the decode rounds of a
    walker could reach if nothing but the demodulator's own instruction stream limited them.  (fraction, file, same_sources)"""
    pd = os.path.join(ROOT, "profiles")
    best = None
    for name in sorted(os.listdir(pd)) if os.path.isdir(pd) else []:
        if "demod_ceiling" not in name or not name.endswith(".json"):
            continue
        try:
            doc = json.load(open(os.path.join(pd, name)))
        except (OSError, ValueError):
            continue
        e = doc.get("cells", {}).get("sf%d-demod%d" % (sf, 0 if demod == 0 else 2))
        if e and e.get("frac_of_hbm_peak"):
            best = (float(e["frac_of_hbm_peak"]), name, doc.get("source_hash") == source_hash())
    return best


# ------------------------------------------------------------------------------------------------- launcher
def respawn_under_launcher(n):
    """`python bench.py --gpus N` on its own: one rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execve(sys.executable, argv, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4])
    ap.add_argument("--path", default="device", choices=["device", "work", "mux"])
    ap.add_argument("--sf", type=int, default=None)
    ap.add_argument("--cr", type=int, default=4)
    ap.add_argument("--samp-rate", type=float, default=1e6, help="config 2 / 3: the decoder's sample rate at BW 125 kHz - 5e5 / 2.5e5 = decimation 4 / 2 (BASELINE's configs are all 1e6: decimation 8)")
    ap.add_argument("--packets", type=int, default=None)
    ap.add_argument("--payload", type=int, default=32)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("LORA_BENCH_STREAMS", "8")))
    ap.add_argument("--seconds", type=float, default=32.0, help="config 4: signal per channel per pass (BASELINE: continuous, at least 2 s; a pass of 8 s leaves the device half empty: 244 jobs of ~1 packet, docs/LAB_NOTEBOOK.md 5.2)")
    ap.add_argument("--demod", type=int, default=2, help="0 grad, 1 fft, 2 fft_compat")
    ap.add_argument("--depth", type=int, default=3, help="pipeline depth: 1 = strictly one pass after the other; 2 = while the device runs "
                    "step k+1 the host stitches step k (decoder handles alternating on one stream; walker kernels never overlap); "
                    "3 = also the envelope pre-pass of step k+2 is issued ahead (it runs in the tail of step k's walker)")
    ap.add_argument("--chunk", type=int, default=1 << 22, help="--path work: items per lora_hip_work call")
    ap.add_argument("--batch", type=int, default=1 << 24, help="--path work: items per device pass (lora_hip_config_t.batch_items)")
    ap.add_argument("--lanes", type=int, default=0, help="independent pass pipelines (each `--depth` handles on a HIP stream of its own) driven by one host thread each; the K "
                    "steps are shared out among them.  A pass of a few hundred jobs (config 4 at 2 s per stream: three dependent launches of 100-160 us with the "
                    "host's planning between them) leaves device AND host waiting on each other; lanes overlap those waits the way a gateway serving several "
                    "antennas would.  0 = the default: 1, config 4 on one GPU: 6 (2 s per stream: 52.6 / 70.6 / 86.3 / 96.7 Gsamples/s with 1 / 2 / 3 / 6 lanes)")
    ap.add_argument("--overlap", action="store_true", help="passes alternate between TWO HIP streams: the walker kernel of pass k+1 starts on the CUs that "
                    "pass k's shorter jobs have left (a streaming receiver's mode; not the default: the per-kernel HIP-event durations then "
                    "include the time a kernel shares the device with its neighbour, and roofline.frac is computed from them)")
    ap.add_argument("--split", action="store_true", help="config 2 / 3 as ONE stream split over the ranks by sample range (SURVEY 8(e), second clause: "
                    "gr_lora_amd.gather.split_stream_ranges - margins on both sides of every cut, frames de-duplicated by header position): strong scaling")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="the K-step block (--steps) is timed repeatedly, each block bracketed by barrier + synchronize on both sides, "
                    "until the timed blocks add up to this many seconds; value / ms_per_step are the MEDIAN block's (0: one block)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grad-line", action="store_true", help="skip the second measurement (the reference's shipped gradient demodulator on the same workload)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_launcher(args.gpus)

    import torch
    import torch.distributed as dist
    from gr_lora_amd import capi, gather

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Under a launcher (WORLD_SIZE set) the process group is RCCL whatever the world size: `torchrun --nproc-per-node 1` drives
    # the same collective code (AsyncSlotGather's all_gather_into_tensor on its side stream, the barrier, the MAX reduction)
    # as an 8-GPU run - tests/test_gpu_rccl.py runs exactly that on the one GPU a gpurun box has.
    use_dist = "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    sf = args.sf if args.sf is not None else (9 if args.config == 4 else 7)
    if args.config == 4:
        channels = gather.shard_streams(8 * world, rank, world)         # 8 channels per GPU: 64 on eight
        cfg, iq, offs, lens, expect = make_gateway_workload(channels, args.seconds, sf)
        wl = "config 4: %d continuous SF%d channels per GPU (%d in all), back-to-back packets of 16-64 B, %.1f s each" % (len(channels), sf, 8 * world, args.seconds)
        wkey = "cfg4-sf%d-%gs" % (sf, args.seconds)
    else:
        packets = args.packets if args.packets is not None else (1024 if args.config == 2 else 256)
        if args.split:
            args.streams = 1
        cfg, iq, offs, lens, expect = make_workload(sf, args.cr, packets, args.payload, min(args.streams, packets), seed=(2 if args.config == 2 else 100 * sf + 4) + (0 if args.split else 1000 * rank), samp_rate=args.samp_rate)
        wl = "SF%d CR4/%d BW125k fs%s, %d packets x %d B payload per GPU, %d stream(s)" % (sf, 4 + args.cr, "1M" if args.samp_rate == 1e6 else "%gk (decimation %d)" % (args.samp_rate / 1e3, cfg.decim),
                                                                                             packets, args.payload, min(args.streams, packets))
        wkey = "cfg%d-sf%d-cr%d-%dx%dB-%dstreams" % (args.config, sf, args.cr, packets, args.payload, min(args.streams, packets)) + ("" if args.samp_rate == 1e6 else "-fs%d" % int(args.samp_rate))
    if args.demod != 2:
        wkey += "-demod%d" % args.demod        # (profiles/*pmc_traffic*.json are keyed by workload AND demodulator: another kernel)
    split_ranges = None
    if args.split and args.config != 4:
        # every rank holds the same capture and decodes its sample range of it (+ margins); frames are owned by header position
        # cuts in the idle gaps (where the serial decoder is in DETECT too): starts of the quiet runs of symbol-long blocks
        eb = (np.abs(iq[:(iq.size // cfg.sps) * cfg.sps].reshape(-1, cfg.sps)[:, ::16]) ** 2).sum(axis=1)
        quiet = eb < 0.25 * np.median(eb[eb > 0]) if np.any(eb > 0) else np.zeros(eb.size, bool)
        gap_cuts = [int(b) * cfg.sps for b in np.flatnonzero(quiet[1:] & ~quiet[:-1]) + 1]
        split_ranges = gather.split_stream_ranges(int(iq.size), world, cfg.sps, max_packet_symbols=8 + synth_payload_symbols(args.payload + 2, cfg), cuts=gap_cuts)
        s0, s1, _lo, _hi = split_ranges[rank]
        whole_items = int(iq.size)
        iq, offs, lens = np.ascontiguousarray(iq[s0:s1]), [0], [s1 - s0]
        wl += "; ONE stream of %d items split over %d rank(s) by sample range" % (whole_items, world)
        wkey += "-split%d" % world
    n_items = int(iq.size)
    n_frames_expected = sum(len(e) for e in expect)
    kw = dict(samp_rate=cfg.samp_rate, bandwidth=cfg.bw, sf=cfg.sf, cr=4, crc=True, reduced_rate=cfg.reduced_rate, device=local_rank, demod=args.demod)

    # What the decoder must publish.  FFT demodulators: the payloads as sent.  Gradient demodulator (--demod 0, the reference's
    # shipped default): that estimator is not the transmitter's inverse on every symbol - the compiled reference itself loses a
    # few payloads of this clean workload - so the yardstick is what THE REFERENCE published on the same IQ:
    # tests/golden/fullsize_ref.json (made by oracle/_ref in the build container; frame count + sha256 per stream).
    ref_fix = None
    # (fixtures exist for the seeds the ranks 0 .. 7 of the default workload use - config2-8streams[-rankR] - and for rank 0's seed of the config-3
    # cells; with --demod 0 a rank without one has no yardstick - the gradient estimator does not reproduce "payloads as sent" even on clean
    # input - and counts as unverified, not as failed: config.verified_ranks, and bit_exact_vs_expected is then null)
    if args.demod == 0 and args.config in (2, 3) and not args.split and args.samp_rate == 1e6:
        try:
            fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_ref.json")))
            want = dict(sf=sf, cr=args.cr, packets=packets, payload=args.payload, streams=min(args.streams, packets), seed=(2 if args.config == 2 else 100 * sf + 4) + 1000 * rank)
            ref_fix = next((e for e in fx.values() if all(e[k] == v for k, v in want.items())), None)
        except (OSError, ValueError):
            ref_fix = None
    unverifiable = args.demod == 0 and args.config in (2, 3) and ref_fix is None and rank != 0 and not args.split
    # --samp-rate (decimation 2 / 4): the reference does not find every packet of these back-to-back workloads there, with any demodulator - "payloads as
    # sent" is no yardstick; what the pass publishes is held to the oracle in tests/test_gpu_decim.py (test_bench_cell_equals_oracle), and the line says null
    no_yardstick = args.samp_rate != 1e6
    unverifiable = unverifiable or no_yardstick or ("LORA_BENCH_NOISE_DB" in os.environ)   # (with noise the reference loses packets: no yardstick either)

    def _digest(frames):
        h = hashlib.sha256()
        for f in frames:
            h.update(len(f).to_bytes(4, "little"))
            h.update(f)
        return h.hexdigest()

    ref_checked = [False]

    def check(frames_by_stream, full=None):
        if unverifiable:
            return True
        if ref_fix is not None and full is not None:
            ref_checked[0] = True
            return all(len(full.get(s, [])) == e["frames"] and _digest(full.get(s, [])) == e["sha256"] for s, e in enumerate(ref_fix["per_stream"]))
        return all(frames_by_stream.get(s, []) == expect[s] for s in range(len(offs)))

    if args.path == "mux":
        res = run_mux_path(args, torch, capi, cfg, iq, offs, lens, kw, expect, wl)
        if rank == 0:
            print(json.dumps(res))
        return
    if args.path == "work":
        res = run_work_path(args, torch, capi, cfg, iq, offs, lens, kw, expect, wl)
        if rank == 0:
            print(json.dumps(res))
        return

    d_iq = torch.from_numpy(iq.view(np.float32)).to(dev)
    depth = max(1, min(3, args.depth))
    hs = [capi.Handle(**kw) for _ in range(depth)]
    stream = torch.cuda.current_stream().cuda_stream
    if args.overlap:
        second = torch.cuda.Stream(device=dev)
        stream = [stream, second.cuda_stream]
    gather_cap = n_frames_expected + 64
    if use_dist:  # all_gather_into_tensor needs one shape on every rank (the gateway workload's packet count differs from rank to rank)
        capt = torch.tensor([gather_cap], dtype=torch.int64, device=dev)
        dist.all_reduce(capt, op=dist.ReduceOp.MAX)
        gather_cap = int(capt.item())
    gat = gather.AsyncSlotGather(dev, gather_cap)

    # One step = one full pass over the batch, software-pipelined the way a streaming receiver runs them
    # (gr_lora_amd.gather.PassPipeline: begin(k+1) before end(k) on one HIP stream, the frames of step k in an asynchronous
    # all_gather that is collected while step k+1 runs).
    pipe = gather.PassPipeline(hs, gat, d_iq.data_ptr(), n_items, offs, lens, stream)
    lanes = args.lanes if args.lanes > 0 else (6 if (args.config == 4 and not use_dist and not args.overlap) else 1)
    if use_dist or args.overlap:
        lanes = 1   # (the ranks' all_gathers must be issued in one order; --overlap is its own experiment)
    if lanes == 1:
        run = pipe.run
    else:
        import threading
        os.environ.setdefault("LORA_HIP_NO_WIDE", "1")   # (two passes in flight share the CUs: the two-per-CU builds of the walker2 kernels, not the 256-register ones a lone launch of <= CUs jobs gets)
        lane_streams = [torch.cuda.Stream(device=dev) for _ in range(lanes - 1)]
        pipes = [pipe] + [gather.PassPipeline([capi.Handle(**kw) for _ in range(depth)], gather.AsyncSlotGather(dev, gather_cap), d_iq.data_ptr(), n_items, offs, lens,
                                              ls.cuda_stream) for ls in lane_streams]
        lock = threading.Lock()

        def run(n_steps, keep=None, check=None):
            """n_steps passes shared out among the lanes (lane i takes every lanes-th step's worth); same return value as PassPipeline.run"""
            share = [n_steps // lanes + (1 if i < n_steps % lanes else 0) for i in range(lanes)]
            out = [None] * lanes
            errs = []

            def locked(fn):
                if fn is None:
                    return None

                def g(done):
                    with lock:
                        fn(done)
                return g

            def work(i):
                try:
                    torch.cuda.set_device(dev)
                    k = [] if keep is not None else None
                    out[i] = pipes[i].run(share[i], k, locked(check))
                    if k:
                        with lock:
                            keep.extend(k)
                except Exception as e:  # noqa: BLE001 - reported by the caller's thread
                    errs.append(e)
            th = [threading.Thread(target=work, args=(i,)) for i in range(1, lanes)]
            for t in th:
                t.start()
            work(0)
            for t in th:
                t.join()
            if errs:
                raise errs[0]
            return sum(o[0] for o in out if o), sum(o[1] for o in out if o)

    # correctness of what is being timed (outside the timed region): frames as gathered, this rank's share, every handle
    kept = []
    run(depth * lanes, kept)
    verified = len(kept) == depth * lanes
    for slots, counts in kept:
        r = rank if len(counts) > 1 else 0
        got, full = {}, {}
        if split_ranges is not None:   # the union of every rank's OWN frames, in stream order, must be the whole capture's frames
            own = []
            for rr in range(len(counts)):
                st_r, _sp, lo, hi = split_ranges[rr]
                own += gather.owned(gather.unpack_frames(slots[rr], counts[rr]), st_r, lo, hi)
            own.sort(key=lambda t: t[2])
            got[0] = [b[15:] for b, _s, _h in own]
            verified = verified and check(got, None)
            continue
        for b, sid, _hp in gather.unpack_frames(slots[r], counts[r]):
            got.setdefault(sid, []).append(b[15:])
            full.setdefault(sid, []).append(b)
        verified = verified and check(got, full)

    # every TIMED step carries its own check: the gathered block of each step (frame counts of every rank + a 64-bit xor over all
    # slot bytes) must equal the block verified frame by frame above - ~15 us of host time per step, inside the timed region
    def fingerprint(done):
        slots, counts = done
        x = 0
        for r, n in enumerate(counts):      # (only the slots in use: what lies behind them is whatever an earlier step left there)
            if n:
                x ^= int(np.bitwise_xor.reduce(np.ascontiguousarray(slots[r, :n]).reshape(-1).view(np.uint64)))
        return (tuple(counts), x)
    want_fp = fingerprint(kept[-1]) if kept else None
    step_fp = {"n": 0, "bad": 0}

    def check_step(done):
        step_fp["n"] += 1
        if want_fp is None or fingerprint(done) != want_fp:
            step_fp["bad"] += 1

    run(40 if n_items < 4e8 else 4)   # pre-roll, untimed like the check above: brings the device to its sustained clocks
    run(args.warmup)                  # the W warm-up steps proper
    # The timed region: EXACTLY --steps steps between barrier + synchronize on both sides, MAX over ranks.  A block of 20 passes of the
    # default workload is 9 ms - too short for anything outside this process to see the device busy - so the block is repeated (every
    # block bracketed the same way, every step of every block fingerprint-checked) until the blocks add up to --min-seconds; the line
    # reports the MEDIAN block (`timed_blocks`, `timed_region_s`, `block_ms_min` / `_max` say what was run).
    def timed_block():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w_ms, n_launch = run(args.steps, check=check_step)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, w_ms, n_launch

    blocks = [timed_block()]
    if args.min_seconds > 0:
        # (the number of blocks must be the same on every rank: it is derived from the first block's MAX-reduced time)
        n_more = min(400, max(0, int(np.ceil(args.min_seconds / max(blocks[0][0], 1e-6))) - 1))
        for _ in range(n_more):
            blocks.append(timed_block())
    order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
    elapsed, walker_ms, launches = blocks[order[len(order) // 2]]
    timed_region_s = float(sum(b[0] for b in blocks))
    steps_timed = args.steps * len(blocks)
    if use_dist:
        tot = torch.tensor([n_items], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_items = whole_items if split_ranges is not None else int(tot.item())   # (split: the capture counts once, not the margins)
        verified = verified and step_fp["bad"] == 0 and step_fp["n"] == steps_timed
        v = torch.tensor([1 if verified else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        verified = bool(v.item())
        nv = torch.tensor([0 if unverifiable else 1], dtype=torch.int64, device=dev)
        dist.all_reduce(nv, op=dist.ReduceOp.SUM)
        verified_ranks = int(nv.item())
    else:
        total_items = whole_items if split_ranges is not None else n_items
        verified = verified and step_fp["bad"] == 0 and step_fp["n"] == steps_timed
        verified_ranks = 0 if unverifiable else 1

    # The same workload through the reference's SHIPPED demodulator (max_frequency_gradient_idx, decoder_impl.cc:499; --demod 0
    # makes it the headline): its own kernels (walker2/3_*_grad), verified against what the compiled reference published on this IQ
    # (tests/golden/fullsize_ref.json).  A second, shorter measurement after the timed region of the headline; single process only.
    grad_line = None
    if rank == 0 and world == 1 and args.config in (2, 3) and args.demod != 0 and not args.no_grad_line and split_ranges is None:
        try:
            fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_ref.json")))
            wantk = dict(sf=sf, cr=args.cr, packets=packets, payload=args.payload, streams=min(args.streams, packets), seed=(2 if args.config == 2 else 100 * sf + 4))
            gfix = next((e for e in fx.values() if all(e[k] == v for k, v in wantk.items())), None) if args.samp_rate == 1e6 else None
        except (OSError, ValueError):
            gfix = None
        ghs = [capi.Handle(**dict(kw, demod=0)) for _ in range(depth)]
        gpipe = gather.PassPipeline(ghs, gather.AsyncSlotGather(dev, gather_cap), d_iq.data_ptr(), n_items, offs, lens, stream)
        gkept = []
        gpipe.run(depth, gkept)
        gver = None
        if gfix is not None:
            gver = len(gkept) == depth
            for slots, counts in gkept:
                full = {}
                for b, sid, _hp in gather.unpack_frames(slots[0], counts[0]):
                    full.setdefault(sid, []).append(b)
                gver = gver and all(len(full.get(k, [])) == e["frames"] and _digest(full.get(k, [])) == e["sha256"] for k, e in enumerate(gfix["per_stream"]))
        gpipe.run(20 if n_items < 4e8 else 2)
        gsteps = max(5, args.steps // 2)
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        gw_ms, _gl = gpipe.run(gsteps)
        torch.cuda.synchronize()
        gel = time.perf_counter() - g0
        gk_ms = gw_ms / gsteps
        grad_line = {"what": "the same workload through the reference's shipped demodulator (max_frequency_gradient_idx, decoder_impl.cc:499): bench.py --demod 0",
                     "value": round(n_items * gsteps / gel / 1e6, 3), "unit": "Msamples/s", "steps": gsteps, "ms_per_step": round(gel / gsteps * 1e3, 4),
                     "kernel": ghs[0].kernel_name(), "kernel_ms_per_pass": round(gk_ms, 4),
                     "frac": round(8.0 * n_items / (gk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if gk_ms > 0 else None,
                     "bit_exact_vs_compiled_reference_frames": gver}
        for hk in ghs:
            hk.close()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_items * args.steps / elapsed / 1e6
        kernel_ms = walker_ms / max(1, args.steps)  # walker kernel time per pass (HIP events, launch stream)
        achieved = 8.0 * n_items / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        kname = hs[0].kernel_name()
        tq = quoted_traffic(wkey)
        frac_events = achieved / HBM_PEAK_GBS
        frac_rocprof = (8.0 * n_items / (tq[2] * 1e-3) / 1e9 / HBM_PEAK_GBS) if tq and tq[2] else None
        ceil = quoted_ceiling(sf, args.demod)
        res = {
            "metric": "IQ Msamples/s demodulated", "value": round(value, 3), "unit": "Msamples/s",
            "symbols_per_s": round(value * 1e6 / cfg.sps, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": ("strong" if split_ranges is not None else "weak"), "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "workload_key": wkey, "items_per_gpu": n_items, "demod": ["grad", "fft", "fft_compat"][args.demod],
                       "bit_exact_vs_expected": (verified if verified_ranks == world else None),   # (null: some rank had no yardstick - --demod 0 beyond rank 0)
                       "verified_steps_in_timed_region": step_fp["n"] - step_fp["bad"], "verified_ranks": verified_ranks,
                       "timed_blocks": len(blocks), "timed_region_s": round(timed_region_s, 4),
                       "block_ms_min": round(blocks[order[0]][0] * 1e3, 4), "block_ms_max": round(blocks[order[-1]][0] * 1e3, 4),
                       "expected": ("frames the compiled reference (oracle/_ref, gradient demodulator) published on this IQ: tests/golden/fullsize_ref.json"
                                    if ref_checked[0] else "payloads as sent"),
                       "parallelism": "streams sharded, dp%d; frame gather: 1 async all_gather per step" % world,
                       "process_group": ("nccl (RCCL), world %d" % world) if use_dist else "none (single process)",
                       "pipeline_depth": depth, "lanes": lanes, "path": "device (IQ resident in HBM)" + (", passes alternating between two HIP streams" if args.overlap else ""),
                       "source_hash": source_hash()},
            # `achieved` / `frac`: THIS run's measurement - the walker kernel's average launch duration from HIP events on the launch stream
            # over the (median) timed block.  `frac_rocprof`: the same kernel's average duration in the committed `rocprofv3 --kernel-trace
            # --stats` summary of this workload, quoted only when workload key and source hash match (another box, another day: a reference
            # point, never the headline).
            # `bound`: the roofline this path is priced against is HBM (8 B per IQ item, no contraction); what actually LIMITS the
            # kernel is its own instruction stream - `valu_ceiling_frac` is the standalone demodulator's measured streaming rate
            # (every window independent, no state machine) over the same peak: the walker cannot exceed it.
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(frac_events, 5),
                         "frac_source": "HIP events of this run on the launch stream (walker kernel, average over the median timed block)",
                         "frac_events": round(frac_events, 5), "frac_rocprof": (round(frac_rocprof, 5) if frac_rocprof is not None else None),
                         "limiter": "VALU issue + round barriers of the state machine, not HBM",
                         "valu_ceiling_frac": (round(ceil[0], 5) if ceil else None),
                         "valu_ceiling_source": (("%s%s" % (ceil[1], "" if ceil[2] else " (measured on other sources of the demodulator)")) if ceil else None),
                         "rocprof_kernel_ms_per_pass": (tq[2] if tq else None), "rocprof_summary": (tq[3] if tq else None),
                         "traffic": tq[0] if tq else None,
                         "traffic_unit": "HBM bytes per pass (rocprofv3 PMC, %s; null unless measured on these sources)" % (tq[1] if tq else "profiles/*pmc_traffic*.json"),
                         "kernel": kname, "kernel_ms_per_pass": round(kernel_ms, 4),
                         "launches_per_pass": launches / max(1, args.steps),
                         "algorithmic_bytes_per_pass": 8 * n_items},
        }
        pp = hs[0].payload_pass()
        if pp["packets"]:
            # a decoupled pass (docs/LAB_NOTEBOOK.md 4.13): `kernel` is the header-only walker variant, `kernel_ms_per_pass` the SUM over the pass's kernels - header-only
            # jobs, the payload pass's symbol and chain kernels, the explicit probes - which overlap (two streams, and the passes of the pipeline among
            # themselves): the job's own fraction of the roofline is value x 8 B / peak
            res["roofline"]["decoupled_pass"] = {"last_pass": pp, "job_frac_of_hbm_peak": round(value * 1e6 * 8.0 / 1e9 / HBM_PEAK_GBS, 5),
                                                 "note": "kernel_ms_per_pass sums kernels that overlap; see docs/LAB_NOTEBOOK.md 4.13 / 5.5"}
        if grad_line is not None:
            res["reference_default_demodulator"] = grad_line
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(cfg, iq, offs, lens)
            res["cpu_baseline"] = {"value": round(cb["grad"][0], 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
                                   "sample": "oracle/lora_oracle.c (C restatement of lib/decoder_impl.cc) built -O3 -march=native on this host, reference-default gradient "
                                             "demodulator, median of %d runs over the first %d items of stream 0; fft demodulator: %.3f Msamples/s" % (cb["grad"][3], cb["grad"][1], cb["fft"][0]),
                                   "reference_build": (None if not cb.get("reference_build") else
                                                       {"value": round(cb["reference_build"][0], 3), "unit": "Msamples/s", "cores": 1,
                                                        "sample": "the reference's lib/decoder_impl.cc itself (oracle/_ref/libref_decoder_fast.so: compiled unmodified, -O3 "
                                                                  "-march=x86-64-v3, VOLK / liquid-dsp as stand-ins of plain loops, its default gradient demodulator), "
                                                                  "same items, median of %d runs" % cb["reference_build"][1]}),
                                   "all_cores": {"value": round(cb["all_cores"][0], 3), "unit": "Msamples/s", "threads": cb["all_cores"][1],
                                                 "host_cores": cb["all_cores"][2], "sample": "one decoder per stream, gradient demod, up to 12e6 items each"}}
        print(json.dumps(res))
    for hk in hs:
        hk.close()
    if lanes > 1:
        for pl in pipes[1:]:
            for hk in pl.hs:
                hk.close()
    if use_dist:
        dist.destroy_process_group()


def run_mux_path(args, torch, capi, cfg, iq, offs, lens, kw, expect_all, wl):
    """--path mux: the gateway as a flowgraph runs it - every stream of the workload is a channel of ONE lora_hip_mux, fed
    round-robin in calls of --chunk items out of page-locked host memory; one device pass per --batch items of every channel.
    Beside it: the same channels through one lora_hip_work handle each (what N decoder blocks amount to without the mux)."""
    nch = len(offs)
    pinned = torch.empty(2 * int(iq.size), dtype=torch.float32, pin_memory=True)
    pinned.numpy()[:] = iq.view(np.float32)
    src = pinned.numpy().view(np.complex64)
    chunk, batch = min(args.chunk, 1 << 18), min(args.batch, 1 << 20)

    def feed(work, drain, flush):
        got = {c: [] for c in range(nch)}
        pos = [0] * nch
        t0 = time.perf_counter()
        live = True
        while live:
            live = False
            for c in range(nch):
                if pos[c] < lens[c]:
                    m = min(chunk, lens[c] - pos[c])
                    work(c, src[offs[c] + pos[c]:offs[c] + pos[c] + m])
                    pos[c] += m
                    live = True
            for b, i, c in drain():
                got[c].append(b[15:])
        flush()
        for b, i, c in drain():
            got[c].append(b[15:])
        return time.perf_counter() - t0, got

    def one_mux():
        m = capi.Mux(nch, batch_items=batch, **kw)
        dt, got = feed(m.work, lambda: [(b, i, i.stream) for b, i in m.drain()], m.flush)
        p = m.passes()
        m.close()
        return dt, got, p[0]

    def one_handles():
        hs = [capi.Handle(batch_items=batch, **kw) for _ in range(nch)]
        def drain():
            out = []
            for c, h in enumerate(hs):
                out += [(b, i, c) for b, i in h.drain()]
            return out
        def flush():
            for h in hs:
                h.flush()
        dt, got = feed(lambda c, a: hs[c].work(a), drain, flush)
        for h in hs:
            h.close()
        return dt, got

    _dt, got, passes = one_mux()
    verified = all(got[c] == expect_all[c] for c in range(nch))
    t_mux = float(np.median([one_mux()[0] for _ in range(max(3, min(args.steps, 5)))]))
    _dt, goth = one_handles()
    verified_h = all(goth[c] == expect_all[c] for c in range(nch))
    t_h = float(np.median([one_handles()[0] for _ in range(3)]))
    n = int(sum(lens))
    return {"metric": "IQ Msamples/s demodulated through lora_hip_mux (host buffers in, PCIe included)", "value": round(n / t_mux / 1e6, 3), "unit": "Msamples/s",
            "n_gpus": 1, "steps": max(3, min(args.steps, 5)), "warmup": 1, "ms_per_step": round(t_mux * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "path": "mux (%d channels through one lora_hip_mux, %d items per call, device passes of %d items per channel)" % (nch, chunk, batch),
                       "device_passes": passes, "bit_exact_vs_expected": verified},
            "one_handle_per_channel": {"value": round(n / t_h / 1e6, 3), "unit": "Msamples/s", "bit_exact_vs_expected": verified_h,
                                       "note": "the same channels and call pattern through %d lora_hip_work handles: one device pass per channel and chunk" % nch}}


def run_work_path(args, torch, capi, cfg, iq, offs, lens, kw, expect_all, wl):
    """--path work: ONE long stream (the workload's streams back to back) fed through lora_hip_work() in calls of --chunk
    items out of page-locked host memory (what a source block's output buffer is to a GNU Radio decoder), device passes
    of --batch items; PCIe included.  The source's own cost (filling its buffer) is not the decoder's and is not timed."""
    n = int(iq.size)
    pinned = torch.empty(2 * n, dtype=torch.float32, pin_memory=True)
    pinned.numpy()[:] = iq.view(np.float32)
    src = pinned.numpy().view(np.complex64)
    expect0 = [f for e in expect_all for f in e]

    def one_pass(collect=False):
        h = capi.Handle(batch_items=args.batch, **kw)
        got, pos = [], 0
        t0 = time.perf_counter()
        while pos < n:
            m = min(args.chunk, n - pos)
            h.work(src[pos:pos + m])
            if collect:
                got += h.drain()
            else:
                h.drain_slots(296)
            pos += m
        h.flush()
        if collect:
            got += h.drain()
        dt = time.perf_counter() - t0
        h.close()
        return dt, got

    _dt, got = one_pass(collect=True)
    verified = [b[15:] for b, _i in got] == expect0
    # measured H2D bandwidth of this box out of the same memory, same call size: the ceiling of this path
    d = torch.empty(args.chunk * 2, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = max(1, min(64, n // args.chunk))
    for r in range(reps):
        d.copy_(pinned[2 * r * args.chunk:2 * (r + 1) * args.chunk], non_blocking=True)
    torch.cuda.synchronize()
    h2d = reps * args.chunk * 8 / (time.perf_counter() - t0) / 1e9
    times = [one_pass()[0] for _ in range(max(3, min(args.steps, 7)))]
    dt = float(np.median(times))
    return {"metric": "IQ Msamples/s demodulated through lora_hip_work (host buffers in, PCIe included)", "value": round(n / dt / 1e6, 3), "unit": "Msamples/s",
            "n_gpus": 1, "steps": len(times), "warmup": 1, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl + "; as ONE stream of %d items" % n, "path": "work (lora_hip_work out of page-locked host memory, %d items per call, device passes of %d items)" % (args.chunk, args.batch),
                       "frames": len(got), "bit_exact_vs_expected": verified},
            "pcie": {"h2d_GBps_measured": round(h2d, 2), "achieved_GBps": round(8 * n / dt / 1e9, 2), "frac_of_h2d": round(8 * n / dt / 1e9 / h2d, 4)}}


if __name__ == "__main__":
    main()
