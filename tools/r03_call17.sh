#!/bin/bash
# round 3, call 17: instruction-cache counters of the SF7 / SF8 / SF9 walkers (code objects of 61 / 80 / 63 KB against a 64 KB instruction cache shared by two CUs)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export LORA_BENCH_CACHE=/dev/shm/lora_bench
for tag in sf7 sf8 sf9; do
  case $tag in sf7) A="";; sf8) A="--config 3 --sf 8 --packets 1024";; sf9) A="--config 3 --sf 9";; esac
  OUT=$REPO/gpurun_out/ic_$tag; rm -rf $OUT; mkdir -p $OUT
  timeout 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $OUT/g1 -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline $A > $OUT/g1.log 2>&1
  echo "$tag rc=$?"; tail -2 $OUT/g1.log | cut -c1-200
  python $REPO/tools/pmc_summary.py $OUT > $REPO/gpurun_out/ic_$tag.json 2>/dev/null
  find $OUT -type f ! -name "*.log" -delete
done
python - <<'P'
import json,os
for t in ("sf7","sf8","sf9"):
    p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out","ic_%s.json"%t)
    try: d=json.load(open(p))
    except Exception as e: print(t,e); continue
    for k,v in d.items():
        if "walker" in k: print(t,k,{c:round(x["mean"]) for c,x in v.items()})
P
