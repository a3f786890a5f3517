#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c4; mkdir -p $O
(timeout 500 python -m pytest tests/test_gpu_zeros.py -m gpu -q -p no:cacheprovider 2>&1 | tail -40) > $O/zeros.txt 2>&1
cat $O/zeros.txt | cut -c1-400
