#!/bin/bash
# round 5, call 1: where the default pass spends its time on this box (per-state clocks, balance, strict SYNC) + first A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c1; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/base.json 2> $O/base.err
LORA_HIP_DEBUG=1 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --min-seconds 0 > $O/debug.json 2> $O/debug.txt
LORA_HIP_STRICT_SYNC=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/fastsync.json 2>/dev/null
LORA_HIP_NO_BALANCE=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/nobalance.json 2>/dev/null
LORA_HIP_JOB_TIMELINE=$PWD/$O/timeline.txt python bench.py --no-cpu-baseline --steps 1 --warmup 0 --min-seconds 0 --depth 1 > /dev/null 2>&1
tail -n 600 $O/timeline.txt > $O/timeline_tail.txt; rm -f $O/timeline.txt
REPS=2 bash tools/ab.sh "--steps 20 --warmup 5" ab/base.so ab/detk2.so > $O/ab.txt 2>&1
for f in base fastsync nobalance; do python -c "import json;d=json.load(open('$O/$f.json'));print('$f',d['value'],d['roofline']['frac'],d['roofline']['kernel_ms_per_pass'],d['config']['bit_exact_vs_expected'],d['config']['timed_blocks'])"; done
cat $O/ab.txt
tail -n 12 $O/debug.txt | cut -c1-300
