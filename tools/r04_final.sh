#!/bin/bash
# round 4, final evidence run on the final sources: the whole GPU suite, smoke(), the demodulator ceiling, then the profile set
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/final_pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
tail -4 gpurun_out/final_pytest.log; tail -2 gpurun_out/final_smoke.log
(cd /tmp && export TMPDIR=/tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/ceil -- python $OLDPWD/tools/demod_ceiling.py run > $OLDPWD/gpurun_out/ceil.log 2>&1)
find gpurun_out/ceil -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
timeout 400 python tools/strict_diag.py > gpurun_out/strict_diag.txt 2>&1
bash tools/profile_all.sh
