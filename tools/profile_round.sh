#!/bin/bash
# One round's profiling evidence for a bench.py workload, written under gpurun_out/prof_<tag>/:
#   stats/   rocprofv3 --kernel-trace --stats (per-kernel durations)
#   fetch/   rocprofv3 --kernel-trace --pmc FETCH_SIZE     (separate passes: the two counters do not fit together,
#   write/   rocprofv3 --kernel-trace --pmc WRITE_SIZE       and the pool forbids PMC with the sys/hip trace domains)
#   line.json  the unprofiled bench line of the same command
# usage: tools/profile_round.sh <tag> [bench.py arguments, e.g. --config 3 --sf 9]; summarise with tools/profile_summary.py <tag> <round>
set -u
TAG=${1:-x}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"   # (on a gpurun box this is a fresh directory; LOCALLY delete gpurun_out/prof_<tag> before merging a new run into it: tools/profile_summary.py takes every file it finds there)
cd /tmp && export TMPDIR=/tmp
STEPS=${PROFILE_STEPS:-10}
CMD="python $REPO/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-grad-line --min-seconds 0 $*"   # (one timed block: the profiler sees STEPS + warm-up + pre-roll passes)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1; echo "stats rc=$?"
CMD="python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-grad-line --min-seconds 0 $*"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1; echo "write rc=$?"
(cd $REPO && timeout 300 python bench.py --steps $STEPS --warmup 2 ${PROFILE_LINE_FLAGS:-} $* 2>/dev/null | tail -1 > "$OUT/line.json")   # the unprofiled line (>= 0.5 s of timed blocks, median); PROFILE_LINE_FLAGS=--no-cpu-baseline: without the CPU legs
tail -1 "$OUT/stats.log" | cut -c1-300
