#!/bin/bash
# A/B of prebuilt library variants on ONE GPU box: tools/ab.sh "<bench args>" ab/x.so ab/y.so ...   (prints walker ms per pass, frac)
args="$1"; shift
cp gr_lora_amd/liblora_hip.so /tmp/keep.so
for rep in 1 2; do
for v in "$@"; do
  cp "$v" gr_lora_amd/liblora_hip.so
  python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_pass'], d['config']['bit_exact_vs_expected'])"
done; done
cp /tmp/keep.so gr_lora_amd/liblora_hip.so
