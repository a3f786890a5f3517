cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c26
export LORA_BENCH_CACHE=/dev/shm/lora_bench
timeout 560 python -m pytest tests/test_gpu_decoupled.py -x -q 2>&1 | tail -25 > gpurun_out/c26/tests.txt
for mode in 0 auto 0 auto; do
  for sec in 2 8; do
  if [ $mode = auto ]; then unset LORA_HIP_DECOUPLED; else export LORA_HIP_DECOUPLED=$mode; fi
  timeout 200 python bench.py --no-cpu-baseline --config 4 --seconds $sec --steps 40 2>gpurun_out/c26/err_$mode.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 ${sec}s decoupled=$mode', d['value'], d['ms_per_step'], d['roofline'].get('frac_events'), d['roofline'].get('kernel_ms_per_pass'), d['config']['bit_exact_vs_expected'], d['roofline'].get('kernel'))" >> gpurun_out/c26/bench.txt 2>&1
  done
done
unset LORA_HIP_DECOUPLED
LORA_HIP_DEBUG=1 timeout 100 python bench.py --no-cpu-baseline --config 4 --seconds 2 --steps 3 --warmup 1 2>&1 | grep -E "payload pass|round1|segment plan|run_jobs host" | tail -12 > gpurun_out/c26/dbg.txt
cat gpurun_out/c26/tests.txt gpurun_out/c26/bench.txt gpurun_out/c26/dbg.txt
