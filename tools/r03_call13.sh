#!/bin/bash
# round 3, call 13: the closed-form fine_sync (FMODE 2): the whole GPU suite, then a same-box A/B against the build without it
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c13
(timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/c13/pytest.log 2>&1
tail -5 gpurun_out/c13/pytest.log
{
echo "## sf7"; REPS=2 tools/ab.sh "" ab/def.so ab/noffs.so
echo "## sf8"; REPS=2 tools/ab.sh "--config 3 --sf 8 --packets 1024" ab/def.so ab/noffs.so
} > gpurun_out/c13/ab.txt 2>&1
cat gpurun_out/c13/ab.txt
