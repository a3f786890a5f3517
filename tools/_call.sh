cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c34
export LORA_BENCH_CACHE=/dev/shm/lora_bench
run() { tag="$1"; shift; for m in 0 auto; do if [ $m = auto ]; then unset LORA_HIP_DECOUPLED; else export LORA_HIP_DECOUPLED=$m; fi; timeout 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag decoupled=$m', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms_per_pass'), d['config']['bit_exact_vs_expected'], d['roofline'].get('kernel'))" >> gpurun_out/c34/bench.txt 2>&1; done; }
run "cfg4 4s" --config 4 --seconds 4 --steps 30 --warmup 3
run "cfg4 8s" --config 4 --seconds 8 --steps 20 --warmup 3
run "cfg4 16s" --config 4 --seconds 16 --steps 12 --warmup 3
run "cfg4 32s" --config 4 --steps 12 --warmup 3
run "cfg3 sf9" --config 3 --sf 9 --steps 30
run "cfg3 sf12" --config 3 --sf 12 --steps 8
run "cfg3 sf9 128" --config 3 --sf 9 --packets 128 --steps 30
run "cfg3 sf9 192" --config 3 --sf 9 --packets 192 --steps 30
run "cfg3 sf12 64" --config 3 --sf 12 --packets 64 --steps 12
run "cfg3 sf11 128" --config 3 --sf 11 --packets 128 --steps 12
cat gpurun_out/c34/bench.txt
