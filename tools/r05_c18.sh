#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c18; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
(LORA_HIP_LIB=$PWD/ab/fcache.so timeout 400 python -m pytest tests/test_gpu_grad_fast.py tests/test_gpu_zeros.py tests/test_gpu_a16.py tests/test_gpu_decoupled.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^  File" | tail -6) > $O/tests.txt 2>&1
cut -c1-250 $O/tests.txt
(LORA_HIP_LIB=$PWD/ab/fcache.so timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "grad_vs_reference and (sf9 or sf10 or sf11 or sf12)" 2>&1 | grep -v "^  File" | tail -4) > $O/tests_full.txt 2>&1
cut -c1-250 $O/tests_full.txt
for sf in 9 10 11 12; do REPS=2 timeout 300 bash tools/ab.sh "--steps 10 --warmup 3 --config 3 --sf $sf --demod 0" ab/head.so ab/fcache.so > $O/ab_sf${sf}g.txt 2>&1; cat $O/ab_sf${sf}g.txt; done
REPS=1 timeout 200 bash tools/ab.sh "--steps 10 --warmup 3 --config 3 --sf 9 --packets 1024 --demod 0" ab/head.so ab/fcache.so > $O/ab_sf9g1024.txt 2>&1; cat $O/ab_sf9g1024.txt
