#!/bin/bash
# round 3, call 22 (exploration for the next round, scratch sources): the SF7 walker at a 256-register budget (one workgroup per CU), without and with the
# worker's next window requested into registers behind its symbol (the first-touch round trip under the round barrier)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c22
REPS=2 tools/ab.sh "" ab/def.so ab/x_eu2.so ab/x_pre256.so > gpurun_out/c22/ab.txt 2>&1
cat gpurun_out/c22/ab.txt
