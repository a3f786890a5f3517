cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c10
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py 2>&1 | tail -5 > gpurun_out/c10/tests.txt
for s in 2 8 32; do for hv in 0 -1; do if [ $hv = 0 ]; then export LORA_HIP_W3_HALF=0; else unset LORA_HIP_W3_HALF; fi; echo "cfg4 seconds=$s half_policy=$hv" >> gpurun_out/c10/ab.txt; timeout 300 python bench.py --config 4 --seconds $s --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac_events'], d['roofline']['kernel'], d['roofline']['kernel_ms_per_pass'], d['roofline']['launches_per_pass'], d['config']['bit_exact_vs_expected'])" >> gpurun_out/c10/ab.txt; done; done
cat gpurun_out/c10/tests.txt gpurun_out/c10/ab.txt
