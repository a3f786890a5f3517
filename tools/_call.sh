cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
timeout 900 python -m pytest tests/test_gpu_strict_sync.py -q 2>&1 | tail -3 > gpurun_out/c5/strict.txt
timeout 900 python tools/strict_diag.py config3-sf7-cr4 config3-sf9-cr4 config3-sf12-cr4 > gpurun_out/c5/diag.txt 2>&1
LORA_HIP_DEBUG=1 timeout 300 python tools/strict_diag.py config3-sf9-cr4 2>&1 | grep -E "per-job avg|^config" > gpurun_out/c5/diag_sf9_debug.txt 2>&1
LORA_HIP_DEBUG=1 timeout 300 python tools/strict_diag.py config3-sf7-cr4 2>&1 | grep -E "per-job avg|^config" > gpurun_out/c5/diag_sf7_debug.txt 2>&1
for s in 1 0 1 0; do LORA_HIP_STRICT_SYNC=$s timeout 300 python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | cut -c1-200 >> gpurun_out/c5/bench_ab.txt; done
cat gpurun_out/c5/strict.txt gpurun_out/c5/diag.txt gpurun_out/c5/diag_sf9_debug.txt gpurun_out/c5/diag_sf7_debug.txt gpurun_out/c5/bench_ab.txt
