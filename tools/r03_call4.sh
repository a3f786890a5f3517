#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c4; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
echo "## sf7 grad" >> $O/ab.txt; REPS=2 tools/ab.sh "--demod 0 --steps 40 --warmup 5" ab/def.so ab/g7eu5.so ab/g7eu6.so >> $O/ab.txt 2>&1
echo "## sf8 grad" >> $O/ab.txt; REPS=2 tools/ab.sh "--config 3 --sf 8 --packets 1024 --demod 0 --steps 30 --warmup 5" ab/def.so ab/g8eu2.so >> $O/ab.txt 2>&1
for sf in 9 10 11 12; do echo "## sf$sf" >> $O/ab.txt; REPS=2 tools/ab.sh "--config 3 --sf $sf --steps 16 --warmup 3" ab/def.so ab/m15.so ab/pf.so >> $O/ab.txt 2>&1; done
cat $O/ab.txt
