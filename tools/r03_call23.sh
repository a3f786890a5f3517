#!/bin/bash
# round 3, call 23 (exploration for the next round, scratch sources): walker3 at SF9 as TWO half-size workgroups per CU (256 threads, 2 windows per round, 79 KB of LDS each)
# against the shipped one (512 threads, 4 windows per round, one per CU), with 1024 and with config 3's 256 packets
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c23
{
echo "## sf9, 1024 packets"; REPS=2 tools/ab.sh "--config 3 --sf 9 --packets 1024 --steps 8 --warmup 2" ab/def.so ab/x_w3half.so
echo "## sf9, 256 packets"; REPS=1 tools/ab.sh "--config 3 --sf 9 --steps 8 --warmup 2" ab/def.so ab/x_w3half.so
} > gpurun_out/c23/ab.txt 2>&1
cat gpurun_out/c23/ab.txt
