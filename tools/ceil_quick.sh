#!/bin/bash
# tools/ceil_quick.sh [sf,sf,...]: the standalone demodulators under rocprofv3 --kernel-trace --stats, kernel averages printed (one GPU call)
sfs=${1:-7,8,9,10,11,12}
R=$PWD
rm -rf gpurun_out/ceilq; mkdir -p gpurun_out/ceilq
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ceilq -- python $R/tools/demod_ceiling.py run $sfs > $R/gpurun_out/ceilq.log 2>&1)
f=$(find gpurun_out/ceilq -name '*kernel_stats.csv' | head -1)
python tools/demod_ceiling.py --collect $f gpurun_out/ceilq.json | python -c "
import sys, json
d = json.load(sys.stdin)
for k in sorted(d, key=lambda s: (int(s.split('-')[0][2:]), s)): print(k, d[k]['avg_ms'], d[k]['frac_of_hbm_peak'])"
