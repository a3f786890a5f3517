cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c7
timeout 1200 python -m pytest tests/test_gpu_strict_sync.py tests/test_gpu_a16.py tests/test_golden.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_determinism.py -q -x 2>&1 | tail -5 > gpurun_out/c7/tests.txt
for sf in 9 10 11 12; do
 for e in 0 1; do
  if [ $e = 1 ]; then export LORA_HIP_NO_EARLY_PROBE=1; else unset LORA_HIP_NO_EARLY_PROBE; fi
  echo "sf$sf no_early=$e" >> gpurun_out/c7/ab.txt
  LORA_HIP_DEBUG=1 timeout 300 python bench.py --config 3 --sf $sf --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/c7/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac_events'], d['roofline']['kernel_ms_per_pass'], d['config']['bit_exact_vs_expected'])" >> gpurun_out/c7/ab.txt
  grep -E "acquisitions|per-job avg|serial fallback" gpurun_out/c7/err.txt | tail -3 >> gpurun_out/c7/ab.txt
 done
done
unset LORA_HIP_NO_EARLY_PROBE
timeout 900 python tools/strict_diag.py config3-sf9-cr4 config3-sf11-cr4 config3-sf12-cr1 > gpurun_out/c7/diag.txt 2>&1
cat gpurun_out/c7/tests.txt gpurun_out/c7/ab.txt gpurun_out/c7/diag.txt
