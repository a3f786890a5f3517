#!/bin/bash
# tools/team_quick.sh [sf,..]: the team FFT demodulator (LORA_HIP_TEAM=1) standalone under rocprofv3, every demod_symbols kernel's average
sfs=${1:-10,11,12}
R=$PWD
rm -rf gpurun_out/ceilq; mkdir -p gpurun_out/ceilq
(cd /tmp && export TMPDIR=/tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ceilq -- python $R/tools/demod_ceiling.py run $sfs > $R/gpurun_out/ceilq.log 2>&1)
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/ceilq/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'demod_symbols' in r['Name'] and 'grad' not in r['Name']:
        avg=float(r['AverageNs'])/1e6
        print(r['Name'].split('(')[0][-44:], r['Calls'], round(avg,4), 'frac', round(8*134217728/(avg*1e-3)/8e12,4))
PY
