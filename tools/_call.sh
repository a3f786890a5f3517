cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c11
timeout 1200 python -m pytest tests/test_gpu_strict_sync.py tests/test_gpu_a16.py tests/test_golden.py tests/test_gpu_flips.py -q -x 2>&1 | tail -4 > gpurun_out/c11/tests.txt
timeout 900 python tools/strict_diag.py config3-sf7-cr4 config3-sf9-cr4 config3-sf10-cr4 config3-sf11-cr4 config3-sf12-cr4 > gpurun_out/c11/diag.txt 2>&1
for s in 1 0 1 0; do LORA_HIP_STRICT_SYNC=$s LORA_HIP_DEBUG=1 timeout 300 python bench.py --steps 60 --no-cpu-baseline 2>gpurun_out/c11/err.txt | cut -c1-200 >> gpurun_out/c11/bench_ab.txt; grep "per-job avg" gpurun_out/c11/err.txt | tail -1 >> gpurun_out/c11/bench_ab.txt; done
cat gpurun_out/c11/tests.txt gpurun_out/c11/diag.txt gpurun_out/c11/bench_ab.txt
