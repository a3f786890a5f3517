#!/usr/bin/env python3
"""Static census of scratch (spill) instructions per device function of lora_kernels.hip (cross-compile, no GPU needed):
   python tools/scratch_census.py [extra hipcc flags, e.g. -DLORA_W3_T512_MASK=15]
Noinline helpers (w3_sync, w3_sfd_round, w2_sync_closed_form) are listed on their own: spills there run once per packet, spills in
a kernel body may sit in the per-symbol rounds."""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "lora_census.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-mllvm", "-greedy-reverse-local-assignment",
                       "-S", "-o", out, os.path.join(root, "gr_lora_amd", "csrc", "lora_kernels.hip")] + sys.argv[1:], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
cur, funcs = None, []
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):\s*(;.*)?$", l)
    if m:
        cur = (m.group(1), i)
    if re.match(r"^\.Lfunc_end\d+:", l) and cur:
        funcs.append((cur[0], cur[1], i)); cur = None
for name, a, b in funcs:
    body = lines[a:b]
    st = sum("scratch_store" in l for l in body); ld = sum("scratch_load" in l for l in body)
    n = sum(bool(re.match(r"^\s+([sv]_|ds_|buffer_|global_|scratch_|flat_)", l)) for l in body)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("lora_hip::", "")
    if "walker" in dn or "w3_" in dn or "w2_" in dn or "demod_symbols" in dn:
        print("%6d instr  scratch st %4d ld %4d  %s" % (n, st, ld, dn))
