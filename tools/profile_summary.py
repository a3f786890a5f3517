#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the files committed under profiles/:
   profiles/<round>_<tag>_kernel_stats.csv   per-kernel stats (rocprofv3 --stats)
   profiles/<round>_<tag>_pmc_traffic.json   HBM bytes per pass of the walker kernel (FETCH_SIZE / WRITE_SIZE), keyed by the
                                             workload and by the hash of the sources it was measured on (bench.py quotes it
                                             as roofline.traffic only when both match)
   profiles/<round>_<tag>_bench_line.json    the unprofiled bench line of the same command
usage: tools/profile_summary.py <tag> <round>"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
src = os.path.join("gpurun_out", "prof_" + tag)
# (gpurun MERGES a run's files into gpurun_out/: files of an earlier run of the same tag may still lie there under other names -
# only the newest file of each kind counts)
def newest(pattern):
    found = glob.glob(pattern, recursive=True)
    return [max(found, key=os.path.getmtime)] if found else []

stats = newest(os.path.join(src, "stats", "**", "*kernel_stats.csv"))
assert stats, "no kernel_stats.csv"
shutil.copy(stats[0], os.path.join("profiles", "%s_%s_kernel_stats.csv" % (rnd, tag)))
# average duration of the walker kernel in that summary, per pass (a pass may launch it more than once: launches_per_pass below)
# (a decoupled pass is three kernels: the header-only walker, the payload pass's symbol kernel and its chain kernel)
PASS_KERNELS = ("walker", "demod_symbols", "payload_chain")
_walker_rows = [r for r in csv.DictReader(open(stats[0])) if any(k in r["Name"] for k in PASS_KERNELS)]

def per_dispatch(kind, counter):
    out = collections.defaultdict(list)  # (kernel, grid) -> values
    for path in newest(os.path.join(src, kind, "**", "*counter_collection.csv")):
        acc, meta = collections.defaultdict(float), {}
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter or not any(k in row["Kernel_Name"] for k in PASS_KERNELS + ("envelope_kernel", "edges_kernel")):
                continue
            acc[row["Dispatch_Id"]] += float(row["Counter_Value"])
            meta[row["Dispatch_Id"]] = (row["Kernel_Name"].split("(")[0], row["Grid_Size"])
        for d, v in acc.items():
            out[meta[d]].append(v)
    return out

fetch, write = per_dispatch("fetch", "FETCH_SIZE"), per_dispatch("write", "WRITE_SIZE")
# the hash bench.py compares is over the sources without comments and white space; it is computed HERE, after checking that the
# box ran exactly these files (the profiled bench line carries the hash it computed there)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
_line0 = json.loads([l for l in open(os.path.join(src, "stats.log")).read().splitlines() if l.startswith("{\"metric\"")][-1])
assert _line0["config"].get("source_hash") in (bench.source_hash(), bench.source_hash(raw=True)), "profile taken on other sources than the working tree"
code_hash = bench.source_hash()
line = json.loads([l for l in open(os.path.join(src, "stats.log")).read().splitlines() if l.startswith("{\"metric\"")][-1])
launches = max(1.0, float(line["roofline"].get("launches_per_pass", 1.0)))
disp = {}
fetch_raw = write_raw = pre_fetch = pre_write = 0.0
walker_keys = [k for k in sorted(set(fetch) | set(write)) if any(p in k[0] for p in PASS_KERNELS)]
for key in sorted(set(fetch) | set(write)):
    f = sum(fetch.get(key, [0])) / max(1, len(fetch.get(key, [])))
    w = sum(write.get(key, [0])) / max(1, len(write.get(key, [])))
    disp["%s grid %s" % key] = {"fetch_kb_avg": f, "write_kb_avg": w, "dispatches_fetch_pass": len(fetch.get(key, []))}
    if any(p in key[0] for p in PASS_KERNELS):   # (a pass may launch the walker more than once - probe jobs - with another grid size: every grid counts once per pass)
        fetch_raw += f * 1024.0
        write_raw += w * 1024.0
    else: # the segment-planning pre-pass (envelope_kernel, edges_kernel)
        pre_fetch += f * 1024.0
        pre_write += w * 1024.0
res = {
    "command": "tools/profile_round.sh %s: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline <workload arguments>" % tag,
    "workload": line["config"]["workload"],
    "workload_key": line["config"].get("workload_key"),
    "source_hash": code_hash,
    "workload_items": line["config"]["items_per_gpu"],
    "units": "FETCH_SIZE / WRITE_SIZE are KB per dispatch (rocprofv3); one pass = one dispatch of each walker grid size listed",
    "dispatches": disp,
    "fetch_bytes_per_pass_raw": fetch_raw,
    "write_bytes_per_pass_raw": write_raw,
    "gfx950_fetch_correction": "x2 (MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies 128-B requests at 64 B); calibrated for 8-byte-per-lane coalesced reads "
                               "(512 contiguous bytes per wave instruction) with tools/calib_fetch.hip: FETCH_SIZE = 0.500 x bytes read; WRITE_SIZE is uncalibrated",
    "hbm_bytes_per_pass_corrected": 2.0 * fetch_raw + write_raw,
    "prepass_hbm_bytes_per_pass_corrected": 2.0 * pre_fetch + pre_write,
    "algorithmic_bytes_per_pass": 8 * line["config"]["items_per_gpu"],
    "traffic_over_algorithmic": (2.0 * fetch_raw + write_raw) / (8.0 * line["config"]["items_per_gpu"]),
    # what bench.py prints as roofline.frac_rocprof / rocprof_kernel_ms_per_pass when workload and sources match
    "rocprof_kernel_stats": "profiles/%s_%s_kernel_stats.csv" % (rnd, tag),
    "rocprof_walker_calls": sum(int(r["Calls"]) for r in _walker_rows),
    "rocprof_walker_avg_ms_per_pass": (sum(float(r["TotalDurationNs"]) for r in _walker_rows) / max(1, sum(int(r["Calls"]) for r in _walker_rows)) * launches / 1e6) if _walker_rows else None,
}
json.dump(res, open(os.path.join("profiles", "%s_%s_pmc_traffic.json" % (rnd, tag)), "w"), indent=1)
lj = os.path.join(src, "line.json")
if os.path.exists(lj) and os.path.getsize(lj) > 10:
    shutil.copy(lj, os.path.join("profiles", "%s_%s_bench_line.json" % (rnd, tag)))
print(json.dumps({k: res[k] for k in ("workload_key", "hbm_bytes_per_pass_corrected", "algorithmic_bytes_per_pass", "traffic_over_algorithmic")}))
