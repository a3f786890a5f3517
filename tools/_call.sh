cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c30
export LORA_BENCH_CACHE=/dev/shm/lora_bench
for ds in 32 48 64 96 64; do
  LORA_HIP_DEC_SEG_SYMBOLS=$ds timeout 100 python bench.py --no-cpu-baseline --config 4 --seconds 2 --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 2s dec seg $ds', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms_per_pass'), d['config']['bit_exact_vs_expected'])" >> gpurun_out/c30/bench.txt
done
LORA_HIP_DEC_SEG_SYMBOLS=32 LORA_HIP_DEBUG=1 timeout 100 python bench.py --no-cpu-baseline --config 4 --seconds 2 --steps 2 --warmup 1 2>&1 | grep -E "payload pass|round1|segment plan|run_jobs host|serial fallback" | tail -6 >> gpurun_out/c30/bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c30/prof -o cfg4 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config 4 --seconds 2 --steps 40 > $GRAFT_REPO_ROOT/gpurun_out/c30/prof_line.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find gpurun_out/c30/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/c30/cfg4_2s_kernel_stats.csv
rm -rf gpurun_out/c30/prof
cat gpurun_out/c30/bench.txt; head -12 gpurun_out/c30/cfg4_2s_kernel_stats.csv | cut -c1-160
