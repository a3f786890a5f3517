#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c9; mkdir -p $O
for a in "7 2 0" "7 2 1" "7 0 0" "8 2 0"; do echo "== $a"; HIP_LAUNCH_BLOCKING=1 AMD_LOG_LEVEL=1 timeout 60 python tools/r05/dbg_z.py $a 2>&1 | grep -v "^  File\|Extension modules" | head -12 | cut -c1-300; done
