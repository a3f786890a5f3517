#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c9; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
for sf in 11 12; do echo "## sf$sf" >> $O/ab.txt; REPS=2 tools/ab.sh "--config 3 --sf $sf --steps 16 --warmup 3" ab/def.so ab/sync2.so >> $O/ab.txt 2>&1; done
cat $O/ab.txt
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_a16.py tests/test_golden.py tests/test_gpu_flips.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
