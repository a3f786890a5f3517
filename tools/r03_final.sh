#!/bin/bash
# round 3, final evidence run on the final sources: the whole GPU suite, smoke(), then the profile set
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/final_pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
tail -4 gpurun_out/final_pytest.log; tail -2 gpurun_out/final_smoke.log
bash tools/profile_all.sh
