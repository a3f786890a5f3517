#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c11; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_detect.py -m gpu -q -p no:cacheprovider 2>&1 | tail -60) > $O/pytest.log 2>&1
tail -25 $O/pytest.log
