#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c8; mkdir -p $O
for k in "2-7" "0-7" "2-9" "0-9" "2-12"; do
  (timeout 120 python -m pytest "tests/test_gpu_zeros.py::test_exact_zero_samples_follow_the_reference[$k]" -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^  File" | head -40) > $O/z_$k.txt 2>&1
  echo "== $k"; head -12 $O/z_$k.txt | cut -c1-250
done
