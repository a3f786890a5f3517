#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c6; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
(timeout 900 python -m pytest tests/test_gpu_mux.py tests/test_gpu_stream_latency.py tests/test_gpu_rccl.py -m gpu -q -p no:cacheprovider 2>&1 | tail -30) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for sf in 9 10 11; do echo "## sf$sf" >> $O/ab.txt; REPS=2 tools/ab.sh "--config 3 --sf $sf --steps 16 --warmup 3" ab/def.so ab/pre7.so >> $O/ab.txt 2>&1; done
cat $O/ab.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/default_line.json
timeout 600 python bench.py --split --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/split_line.json
cut -c1-1500 $O/default_line.json; cut -c1-700 $O/split_line.json
