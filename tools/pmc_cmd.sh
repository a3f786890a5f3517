#!/bin/bash
# SQ counters for an arbitrary command: one rocprofv3 --pmc pass per counter group (kernel-trace only, as the pool
# requires), CSV under gpurun_out/pmc_<tag>/.  usage: tools/pmc_cmd.sh <tag> <command...>; summarise with tools/pmc_summary.py
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  (cd $REPO && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -- "$@" > "$OUT/g$i.log" 2>&1)
  echo "group $i rc=$? : $grp"
done
