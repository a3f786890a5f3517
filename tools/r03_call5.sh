#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c5; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
(time timeout 1200 python -m pytest tests/test_gpu_mux.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_a16.py -m gpu -q -p no:cacheprovider 2>&1 | tail -80) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 600 python bench.py --path mux --config 4 --seconds 2 --steps 5 2>$O/mux.err | tail -1 > $O/mux_cfg4_2s.json
timeout 600 python bench.py --path mux --config 2 --steps 5 2>>$O/mux.err | tail -1 > $O/mux_cfg2.json
for s in 2 8 32; do timeout 600 python bench.py --config 4 --seconds $s --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/cfg4.jsonl; done
cat $O/mux_cfg4_2s.json $O/mux_cfg2.json $O/cfg4.jsonl | cut -c1-600
