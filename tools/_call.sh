cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c37
export LORA_BENCH_CACHE=/dev/shm/lora_bench
prof() { tag=$1; shift; (cd /tmp && export TMPDIR=/tmp && env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c37/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config 4 --seconds 2 --steps 30 > /dev/null 2>&1); find gpurun_out/c37/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/c37/$tag.csv; rm -rf gpurun_out/c37/prof_$tag; echo "== $tag"; python - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/c37/$tag.csv')):
    if any(k in r['Name'] for k in ('walker','demod','payload')): print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
}
prof default A=1
prof nosecond LORA_HIP_NO_SECOND_READS=1
prof grad A=1 LORA_BENCH_DEMOD0=1
