#!/bin/bash
# round 3, call 18: the gradient demodulator's profile set for the SFs profile_all.sh leaves out (same sources as the committed set)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LORA_BENCH_CACHE=/dev/shm/lora_bench
mkdir -p gpurun_out
{
tools/profile_round.sh sf8_grad --config 3 --sf 8 --packets 1024 --demod 0
PROFILE_STEPS=8 tools/profile_round.sh sf10_grad --config 3 --sf 10 --demod 0
PROFILE_STEPS=6 tools/profile_round.sh sf11_grad --config 3 --sf 11 --demod 0
} > gpurun_out/c18.log 2>&1
find gpurun_out/prof_* -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.log" ! -name "line.json" -delete 2>/dev/null
tail -5 gpurun_out/c18.log | cut -c1-200
