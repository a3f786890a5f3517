#!/bin/bash
# tools/walk_quick.sh "<sf:packets> ...": bench.py lines (FFT walker only, no CPU baseline) for config-3 style cells on ONE box; prints value, frac, kernel ms, bit-exactness
export LORA_BENCH_CACHE=${LORA_BENCH_CACHE:-/dev/shm/lora_bench}
for cell in ${1:-7:1024 8:1024 9:256 10:256 11:256 12:256}; do
  sf=${cell%%:*}; pk=${cell##*:}
  if [ "$sf" = "7" ] && [ "$pk" = "1024" ]; then args=""; else args="--config 3 --sf $sf --packets $pk"; fi
  python bench.py --no-cpu-baseline --no-grad-line $args ${EXTRA} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('sf$sf x $pk:', d['value'], 'Msamples/s  frac', r['frac'], ' kernel ms', r.get('kernel_ms_per_pass'), ' traffic', r.get('traffic'), ' exact', d['config'].get('bit_exact_vs_expected'), r.get('kernel'))"
done
