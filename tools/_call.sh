cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c40
export LORA_BENCH_CACHE=/dev/shm/lora_bench
timeout 300 python -m pytest tests/test_gpu_strict_sync.py tests/test_golden.py tests/test_gpu_a16.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/c40/tests1.txt
timeout 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "grad_vs_reference_fixture and (sf9 or sf10 or sf11 or sf12)" 2>&1 | tail -4 > gpurun_out/c40/tests2.txt
cp gr_lora_amd/liblora_hip.so ab/par.so
{
echo "## sf9";  REPS=2 bash tools/ab.sh "--config 3 --sf 9 --steps 30" ab/def.so ab/par.so
echo "## sf12"; REPS=1 bash tools/ab.sh "--config 3 --sf 12 --steps 8" ab/def.so ab/par.so
echo "## sf11"; REPS=1 bash tools/ab.sh "--config 3 --sf 11 --steps 12" ab/def.so ab/par.so
} > gpurun_out/c40/ab.txt 2>&1
cat gpurun_out/c40/tests1.txt gpurun_out/c40/tests2.txt gpurun_out/c40/ab.txt
