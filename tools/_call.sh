cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c39
export LORA_BENCH_CACHE=/dev/shm/lora_bench
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCC_[A-Z0-9_]*(DRAM|MALL|IO|GMI|RDREQ|WRREQ)[A-Z0-9_]*|MALL[A-Z0-9_]*|[A-Z0-9_]*HBM[A-Z0-9_]*)\b" | sort -u > gpurun_out/c39/counters.txt
wc -l gpurun_out/c39/counters.txt; head -60 gpurun_out/c39/counters.txt | tr '\n' ' '
cd /tmp && export TMPDIR=/tmp
for c in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c39/$c -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --config 3 --sf 12 > $GRAFT_REPO_ROOT/gpurun_out/c39/$c.log 2>&1; echo "$c rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
for c in ("TCC_EA0_RDREQ_sum","TCC_EA0_RDREQ_DRAM_sum","TCC_EA0_RDREQ_32B_sum","TCC_BUBBLE_sum"):
    fs=glob.glob("gpurun_out/c39/%s/**/*counter_collection.csv"%c, recursive=True)
    if not fs: print(c,"no file"); continue
    acc=collections.defaultdict(float); name={}
    for r in csv.DictReader(open(fs[0])):
        if "walker3_kernel_sf12" in r["Kernel_Name"] and r["Counter_Name"]==c: acc[r["Dispatch_Id"]]+=float(r["Counter_Value"])
    v=sorted(acc.values()); print(c, "dispatches", len(v), "median per dispatch", v[len(v)//2] if v else None)
PY
find gpurun_out/c39 -type f ! -name "*.txt" ! -name "*.log" -delete
