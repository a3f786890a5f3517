#!/bin/bash
# One round's profiling evidence for bench.py's workload, written under gpurun_out/prof_<tag>/:
#   stats/   rocprofv3 --kernel-trace --stats (per-kernel durations)
#   fetch/   rocprofv3 --kernel-trace --pmc FETCH_SIZE     (separate passes: the two counters do not fit together,
#   write/   rocprofv3 --kernel-trace --pmc WRITE_SIZE       and the pool forbids PMC with the sys/hip trace domains)
# Summarise with tools/profile_summary.py <tag>.
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1; echo "stats rc=$?"
CMD="python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1; echo "write rc=$?"
tail -1 "$OUT/stats.log"
