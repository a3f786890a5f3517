#!/bin/bash
# tools/dbg_states.sh "<bench args>" lib.so ...: LORA_HIP_DEBUG per-state clocks of the walker for library variants on one box
args="$1"; shift
export LORA_BENCH_CACHE=${LORA_BENCH_CACHE:-/dev/shm/lora_bench}
for v in "$@"; do
  echo "== $v"
  LORA_HIP_LIB=$PWD/$v LORA_HIP_DEBUG=1 python bench.py --no-cpu-baseline --no-grad-line --steps 3 --warmup 1 --min-seconds 0 $args 2>&1 | grep -E "per-job avg|control per job|job kcycles|walker 0" | tail -4
done
