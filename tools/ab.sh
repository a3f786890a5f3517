#!/bin/bash
# A/B of prebuilt library variants on ONE GPU box: tools/ab.sh "<bench args>" ab/x.so ab/y.so ...   (prints whole-job Msamples/s,
# frac, walker ms per pass, bit-exactness; REPS rounds, variants interleaved; the workload is synthesised once)
args="$1"; shift
export LORA_BENCH_CACHE=${LORA_BENCH_CACHE:-/dev/shm/lora_bench}
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  LORA_HIP_LIB=$PWD/$v python bench.py --no-cpu-baseline --no-grad-line $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_pass'], d['config']['bit_exact_vs_expected'])"
done; done
