#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c15; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
(timeout 300 python -m pytest tests/test_gpu_decoupled.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^  File" | tail -25) > $O/dec.txt 2>&1
tail -25 $O/dec.txt | cut -c1-300
(time timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_decoupled.py 2>&1 | grep -v "^  File" | tail -15) > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt | cut -c1-300
timeout 100 python bench.py --no-cpu-baseline --steps 20 --warmup 5 | python -c "import sys,json;d=json.loads(sys.stdin.read());print(d['value'],d['roofline']['frac'],d['config']['bit_exact_vs_expected'])"
timeout 100 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --config 3 --sf 7 --packets 64 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('sf7 64 packets',d['value'],d['roofline']['kernel'],d['config']['bit_exact_vs_expected'])"
LORA_HIP_DECOUPLED=0 timeout 100 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --config 3 --sf 7 --packets 64 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('sf7 64 packets ordinary',d['value'],d['roofline']['kernel'],d['config']['bit_exact_vs_expected'])"
