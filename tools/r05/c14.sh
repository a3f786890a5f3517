#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c14; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
(LORA_HIP_LIB=$PWD/ab/tsfd.so timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_flips.py tests/test_gpu_zeros.py tests/test_gpu_determinism.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^  File" | tail -8) > $O/tests.txt 2>&1
(LORA_HIP_LIB=$PWD/ab/tsfd.so timeout 200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "config2 or sf7 or sf8" 2>&1 | tail -4) > $O/tests_full.txt 2>&1
REPS=3 timeout 300 bash tools/ab.sh "--steps 20 --warmup 5" ab/head.so ab/tsfd.so > $O/ab.txt 2>&1
REPS=2 timeout 200 bash tools/ab.sh "--steps 20 --warmup 5 --demod 0" ab/head.so ab/tsfd.so > $O/ab_grad.txt 2>&1
REPS=1 timeout 200 bash tools/ab.sh "--steps 20 --warmup 5 --config 3 --sf 8 --packets 1024" ab/head.so ab/tsfd.so > $O/ab_sf8.txt 2>&1
LORA_HIP_LIB=$PWD/ab/tsfd.so LORA_HIP_DEBUG=1 timeout 100 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --min-seconds 0 > /dev/null 2> $O/debug.txt
cat $O/tests.txt $O/tests_full.txt $O/ab.txt $O/ab_grad.txt $O/ab_sf8.txt; grep -E "acquisitions|per-job avg|round2" $O/debug.txt | tail -3 | cut -c1-260
