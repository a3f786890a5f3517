cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c24
timeout 300 python tools/strict_diag.py config3-sf12-cr1 config3-sf11-cr1 config3-sf11-cr2 config3-sf11-cr3 > gpurun_out/c24/diag.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -k "grad_vs_reference_fixture" 2>&1 | tail -3 > gpurun_out/c24/tests.txt
cat gpurun_out/c24/tests.txt gpurun_out/c24/diag.txt
