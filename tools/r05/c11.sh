#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c11; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
(LORA_HIP_LIB=$PWD/ab/zm3inl.so timeout 300 python -m pytest tests/test_gpu_zeros.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^  File" | tail -12) > $O/zeros_inl.txt 2>&1
(LORA_HIP_LIB=$PWD/ab/zm3.so timeout 300 python -m pytest tests/test_gpu_zeros.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^  File" | tail -12) > $O/zeros_noinl.txt 2>&1
for sf in 9 10 11 12; do REPS=2 bash tools/ab.sh "--steps 10 --warmup 3 --config 3 --sf $sf" ab/base.so ab/zm3.so ab/zm3inl.so > $O/ab_sf$sf.txt 2>&1; done
for sf in 9 10 12; do REPS=1 bash tools/ab.sh "--steps 10 --warmup 3 --config 3 --sf $sf --demod 0" ab/base.so ab/zm3.so ab/zm3inl.so > $O/ab_sf${sf}g.txt 2>&1; done
cut -c1-200 $O/zeros_inl.txt; cut -c1-200 $O/zeros_noinl.txt; cat $O/ab_sf9.txt $O/ab_sf10.txt $O/ab_sf11.txt $O/ab_sf12.txt $O/ab_sf9g.txt $O/ab_sf10g.txt $O/ab_sf12g.txt
