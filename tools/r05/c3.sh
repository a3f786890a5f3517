#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c3; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
LORA_HIP_LIB=$PWD/ab/diag.so LORA_HIP_DEBUG=1 LORA_HIP_DEBUG_DEFER=1 python bench.py --no-cpu-baseline --steps 1 --warmup 0 --min-seconds 0 --depth 1 > $O/debug.json 2> $O/debug.txt
grep "defer" $O/debug.txt | head -14
