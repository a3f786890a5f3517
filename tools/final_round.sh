#!/bin/bash
# The round's final evidence run on the final sources, in two parts (each one gpurun call, each step under its own timeout):
#   tools/final_round.sh tests      the whole GPU suite, smoke(), the standalone demodulator ceilings under rocprofv3
#   tools/final_round.sh profiles   tools/profile_all.sh: per-SF kernel stats + PMC traffic + bench lines, the default / work / config-4 / RCCL lines, SQ counters
# afterwards, here: for t in <tags>; do python tools/profile_summary.py $t rNN; done; python tools/demod_ceiling.py --collect ...; python tools/collect_round.py rNN
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
case "${1:-tests}" in
tests)
  (time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^  File" | tail -40) > gpurun_out/final_pytest.log 2>&1
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
  tail -4 gpurun_out/final_pytest.log; tail -2 gpurun_out/final_smoke.log
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/ceil -- python $OLDPWD/tools/demod_ceiling.py run > $OLDPWD/gpurun_out/ceil.log 2>&1)
  find gpurun_out/ceil -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
  tail -3 gpurun_out/ceil.log
  ;;
profiles)
  timeout 3300 bash tools/profile_all.sh
  ;;
esac
