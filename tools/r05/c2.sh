#!/bin/bash
# round 5, call 2: deferred SYNC ties (correctness on the SF7/SF8 suites, hit rate) and the FCFS priority policy, A/B on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c2; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
(LORA_HIP_LIB=$PWD/ab/defer.so timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_strict_sync.py tests/test_gpu_edges.py tests/test_gpu_determinism.py tests/test_gpu_a16.py tests/test_golden.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8) > $O/tests.txt 2>&1
(LORA_HIP_LIB=$PWD/ab/defer.so timeout 300 python -m pytest "tests/test_gpu_fullsize.py" -m gpu -q -x -p no:cacheprovider -k "config2 or sf7 or sf8" 2>&1 | tail -8) > $O/tests_full.txt 2>&1
LORA_HIP_LIB=$PWD/ab/defer.so LORA_HIP_DEBUG=1 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --min-seconds 0 > $O/debug.json 2> $O/debug.txt
REPS=2 bash tools/ab.sh "--steps 20 --warmup 5" ab/base.so ab/defer.so ab/fcfs0.so ab/fcfs1.so ab/fcfs2.so > $O/ab.txt 2>&1
REPS=1 bash tools/ab.sh "--steps 20 --warmup 5 --demod 0" ab/base.so ab/defer.so ab/fcfs1.so > $O/ab_grad.txt 2>&1
REPS=1 bash tools/ab.sh "--steps 20 --warmup 5 --config 3 --sf 8 --packets 1024" ab/base.so ab/defer.so ab/fcfs1.so > $O/ab_sf8.txt 2>&1
cat $O/tests.txt $O/tests_full.txt $O/ab.txt $O/ab_grad.txt $O/ab_sf8.txt
grep -E "near-ties|per-job avg" $O/debug.txt | tail -4 | cut -c1-250
