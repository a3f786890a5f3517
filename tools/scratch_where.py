#!/usr/bin/env python3
"""Where the spill instructions of one kernel come from: scratch_store / scratch_load counts per source line (.loc), cross-compile only.
   python tools/scratch_where.py <kernel substring> [extra hipcc flags]"""
import collections, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "lora_where.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-mllvm", "-greedy-reverse-local-assignment", "-gline-tables-only",
                       "-S", "-o", out, os.path.join(root, "gr_lora_amd", "csrc", "lora_kernels.hip")] + sys.argv[2:], stderr=subprocess.DEVNULL)
files, cur, infn, loc = {}, None, False, ("?", 0)
st, ld = collections.Counter(), collections.Counter()
for l in open(out):
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    m = re.match(r"^(_Z\w+):", l)
    if m:
        infn = sys.argv[1] in m.group(1)
    if l.startswith(".Lfunc_end"):
        infn = False
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
    if infn and "scratch_store" in l:
        st[loc] += 1
    if infn and "scratch_load" in l:
        ld[loc] += 1
for k in sorted(set(st) | set(ld), key=lambda k: -(st[k] + ld[k]))[:40]:
    print("%-28s line %5d   st %3d  ld %3d" % (k[0], k[1], st[k], ld[k]))
