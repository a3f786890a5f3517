#!/bin/bash
# round 4, after the last kernel changes (payload chain kernel, two DETECT windows per group in the header-only variants): the whole GPU suite again, then
# the profile sets whose kernels or launch paths changed - and the headline's, so that bench.py finds a rocprofv3 summary taken on the final sources
cd ${GRAFT_REPO_ROOT:-/root/repo}
export LORA_BENCH_CACHE=/dev/shm/lora_bench
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/final2_pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final2_smoke.log 2>&1
tail -4 gpurun_out/final2_pytest.log; tail -1 gpurun_out/final2_smoke.log
{
tools/profile_round.sh sf7
PROFILE_STEPS=8 tools/profile_round.sh sf9 --config 3 --sf 9
PROFILE_STEPS=4 tools/profile_round.sh sf12 --config 3 --sf 12
PROFILE_STEPS=20 tools/profile_round.sh cfg4_2s --config 4 --seconds 2
python bench.py 2>/dev/null | tail -1 > gpurun_out/default_line.json
LORA_HIP_STRICT_SYNC=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/default_fast_sync_line.json
python bench.py --demod 0 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/default_grad_line.json
python bench.py --config 4 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg4_line.json
python bench.py --config 4 --seconds 8 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg4_8s_line.json
python bench.py --config 4 --seconds 2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg4_2s_line.json
LORA_HIP_DECOUPLED=0 python bench.py --config 4 --seconds 2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg4_2s_ordinary_line.json
} > gpurun_out/final2_profile.log 2>&1
find gpurun_out/prof_* -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.log" ! -name "line.json" -delete 2>/dev/null
tail -5 gpurun_out/final2_profile.log
