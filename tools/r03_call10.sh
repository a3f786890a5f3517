#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c10; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
(timeout 900 python -m pytest tests/test_gpu_detect.py tests/test_gpu_mux.py -m gpu -q -p no:cacheprovider 2>&1 | tail -60) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 900 python tools/stress_r03.py 20 > $O/stress.txt 2>&1
tail -40 $O/stress.txt
