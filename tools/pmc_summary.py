#!/usr/bin/env python3
"""Summarises gpurun_out/pmc/g*/**/*counter_collection.csv: per kernel name, mean of every counter per dispatch."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(root + "/g*/**/*counter_collection.csv", recursive=True):
    per_dispatch = collections.defaultdict(float)
    names = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = row["Kernel_Name"].split("(")[0]
    for (disp, ctr), v in per_dispatch.items():
        acc[names[disp]][ctr].append(v)
out = {}
for k, ctrs in acc.items():
    out[k] = {c: {"mean": sum(v) / len(v), "max": max(v), "n": len(v)} for c, v in sorted(ctrs.items())}
json.dump(out, sys.stdout, indent=1)
