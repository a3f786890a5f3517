#!/bin/bash
# round 3, call 16: the stress set once more on the final build (determinism of repeated passes, many streams, every kernel family)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c16; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
timeout 300 python tools/stress_determinism.py 200 > $O/det_fft.txt 2>&1; tail -2 $O/det_fft.txt
timeout 300 python tools/stress_streams.py 60 1024 > $O/streams.txt 2>&1; tail -2 $O/streams.txt
timeout 600 python tools/stress_r03.py 60 > $O/stress.txt 2>&1; tail -3 $O/stress.txt
