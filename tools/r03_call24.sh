#!/bin/bash
# round 3, call 24 (exploration, scratch sources): the half-size walker3 workgroups at SF10 (T = 256, NG = 1, half twiddle table: 77 984 B of LDS, two per CU)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c24
{
echo "## sf10, 1024 packets"; REPS=2 tools/ab.sh "--config 3 --sf 10 --packets 1024 --steps 6 --warmup 2" ab/def.so ab/x_w3half10.so
echo "## sf10, 256 packets"; REPS=1 tools/ab.sh "--config 3 --sf 10 --steps 6 --warmup 2" ab/def.so ab/x_w3half10.so
} > gpurun_out/c24/ab.txt 2>&1
cat gpurun_out/c24/ab.txt
