#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c12; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
timeout 600 python tools/stress_determinism.py 300 > $O/det_fft.txt 2>&1; tail -2 $O/det_fft.txt
timeout 600 python tools/stress_streams.py 100 1024 > $O/streams.txt 2>&1; tail -2 $O/streams.txt
timeout 900 python tools/stress_r03.py 120 > $O/stress.txt 2>&1; tail -3 $O/stress.txt
