#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c8; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
for sf in 9 10 11 12; do echo "## sf$sf" >> $O/ab.txt; REPS=2 tools/ab.sh "--config 3 --sf $sf --steps 16 --warmup 3" ab/def.so ab/p1.so >> $O/ab.txt 2>&1; done
cat $O/ab.txt
LORA_HIP_LIB=$PWD/ab/p1.so LORA_HIP_W3_STAMPS=1 timeout 300 python tools/demod_bench.py 9 > $O/stamps_p1_sf9.txt 2>&1
grep -A9 "w3 demod phases" $O/stamps_p1_sf9.txt | tail -10
