#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of an arbitrary command (separate passes), CSV under gpurun_out/traffic_<tag>/; prints per-kernel means
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$REPO/gpurun_out/traffic_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd $REPO && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- "$@" > "$OUT/$c.log" 2>&1)
done
cd $REPO && python - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); names = {}
    for p in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == c:
                acc[r["Dispatch_Id"]][c] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
    by = collections.defaultdict(list)
    for d, v in acc.items(): by[names[d]].append(v[c])
    for k, v in by.items(): print(c, k[:60], "n=%d mean_KB=%.0f" % (len(v), sum(v) / len(v)))
PY
