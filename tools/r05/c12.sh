#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c12; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
for sf in 9 11; do REPS=2 bash tools/ab.sh "--steps 10 --warmup 3 --config 3 --sf $sf" ab/base.so ab/zm3.so ab/zm3noz.so > $O/ab_sf$sf.txt 2>&1; done
REPS=2 bash tools/ab.sh "--steps 10 --warmup 3 --config 3 --sf 10 --demod 0" ab/base.so ab/zm3.so ab/zm3noz.so > $O/ab_sf10g.txt 2>&1
cat $O/ab_sf9.txt $O/ab_sf11.txt $O/ab_sf10g.txt
