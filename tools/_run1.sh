mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu.log 2>&1; echo "tests rc=$?" 
tail -4 gpurun_out/t_gpu.log
for d in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --steps 60 --depth $d 2>gpurun_out/bench_d$d.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('depth $d', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_pass'], d['roofline']['frac'], d['config']['bit_exact_vs_expected'])"
done
