#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c6; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
REPS=2 bash tools/ab.sh "--steps 20 --warmup 5" ab/base.so ab/zfix.so ab/znoslow.so > $O/ab.txt 2>&1
REPS=2 bash tools/ab.sh "--steps 20 --warmup 5 --demod 0" ab/base.so ab/zfix.so ab/znoslow.so > $O/ab_grad.txt 2>&1
for sf in 9 10; do REPS=1 bash tools/ab.sh "--steps 10 --warmup 3 --config 3 --sf $sf" ab/base.so ab/zfix.so ab/znoslow.so > $O/ab_sf$sf.txt 2>&1; done
cat $O/ab.txt $O/ab_grad.txt $O/ab_sf9.txt $O/ab_sf10.txt
