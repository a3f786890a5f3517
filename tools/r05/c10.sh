#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LORA_BENCH_CACHE=/dev/shm/lora_bench
for a in "7 2 0" "7 0 0" "8 2 1"; do echo "== $a"; LORA_HIP_LIB=$PWD/ab/zminl.so timeout 60 python tools/r05/dbg_z.py $a 2>&1 | grep -v "^  File\|Extension modules" | head -6 | cut -c1-200; done
REPS=2 bash tools/ab.sh "--steps 20 --warmup 5" ab/base.so ab/zminl.so
REPS=1 bash tools/ab.sh "--steps 20 --warmup 5 --demod 0" ab/base.so ab/zminl.so
