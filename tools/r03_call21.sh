#!/bin/bash
# round 3, call 21: what one more instruction per symbol costs in the SF7 walker, by class - variants of the shipped library with 200 padding instructions in front of the
# dechirp (v_nop / s_nop 0 / a dependent s_and_b32 chain), built from a scratch copy of the sources (the tree's sources do not carry the switches)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c21
REPS=3 tools/ab.sh "" ab/def.so ab/pad_vnop.so ab/pad_snop.so ab/pad_salu.so > gpurun_out/c21/ab.txt 2>&1
cat gpurun_out/c21/ab.txt
