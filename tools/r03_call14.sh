#!/bin/bash
# round 3, call 14: same-box A/B of the closed-form fine_sync against the build without it
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c14
{
echo "## sf7"; REPS=3 tools/ab.sh "" ab/def.so ab/noffs.so
echo "## sf8"; REPS=2 tools/ab.sh "--config 3 --sf 8 --packets 1024" ab/def.so ab/noffs.so
} > gpurun_out/c14/ab.txt 2>&1
cat gpurun_out/c14/ab.txt
