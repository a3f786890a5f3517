#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c7; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
(timeout 500 python -m pytest tests/test_gpu_zeros.py -m gpu -q -p no:cacheprovider 2>&1 | tail -30) > $O/zeros.txt 2>&1
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_a16.py tests/test_gpu_strict_sync.py tests/test_golden.py tests/test_gpu_grad_fast.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8) > $O/tests.txt 2>&1
REPS=2 bash tools/ab.sh "--steps 20 --warmup 5" ab/base.so ab/zm.so > $O/ab.txt 2>&1
REPS=2 bash tools/ab.sh "--steps 20 --warmup 5 --demod 0" ab/base.so ab/zm.so > $O/ab_grad.txt 2>&1
for sf in 9 10 12; do REPS=1 bash tools/ab.sh "--steps 10 --warmup 3 --config 3 --sf $sf" ab/base.so ab/zm.so > $O/ab_sf$sf.txt 2>&1; done
REPS=1 bash tools/ab.sh "--steps 10 --warmup 3 --config 3 --sf 10 --demod 0" ab/base.so ab/zm.so > $O/ab_sf10g.txt 2>&1
cat $O/zeros.txt | cut -c1-250; cat $O/tests.txt $O/ab.txt $O/ab_grad.txt $O/ab_sf9.txt $O/ab_sf10.txt $O/ab_sf12.txt $O/ab_sf10g.txt
