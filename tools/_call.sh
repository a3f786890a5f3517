cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c12
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -k "grad_vs_reference_fixture" 2>&1 | tail -5 > gpurun_out/c12/fullsize_grad.txt
timeout 900 python -m pytest tests/test_gpu_channelizer.py tests/test_gpu_strict_sync.py tests/test_gpu_parity.py -q 2>&1 | tail -3 > gpurun_out/c12/tests.txt
timeout 600 python tools/strict_diag.py config3-sf11-cr1 config3-sf12-cr1 config3-sf12-cr2 config3-sf11-cr3 > gpurun_out/c12/diag.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c12/ceil -- python $GRAFT_REPO_ROOT/tools/demod_ceiling.py run > $GRAFT_REPO_ROOT/gpurun_out/c12/ceil.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/c12/ceil -type f ! -name "*kernel_stats.csv" -delete
cat gpurun_out/c12/fullsize_grad.txt gpurun_out/c12/tests.txt gpurun_out/c12/diag.txt; find gpurun_out/c12/ceil -name "*kernel_stats.csv" | head -2; tail -3 gpurun_out/c12/ceil.log
