mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu.log 2>&1; echo "tests rc=$?" 
tail -4 gpurun_out/t_gpu.log
LORA_HIP_DEBUG=1 timeout 300 python tools/seg_stats.py 0 2> gpurun_out/seg0.log; grep -A12 "pass 2" gpurun_out/seg0.log | grep -E "per-job|kcycles over|plan "
timeout 300 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/bench_plan.json 2>gpurun_out/bench_plan.err; python -c "import json; d=json.load(open('gpurun_out/bench_plan.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_pass'], d['config']['bit_exact_vs_expected'])"
