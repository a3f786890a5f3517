#!/bin/bash
# Collects SQ counters for the walker kernel: one rocprofv3 --pmc pass per counter group (kernel-trace only,
# as the pool requires), CSV output under gpurun_out/pmc/<group>/.  Summarise with tools/pmc_summary.py.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
# usage: tools/pmc_walker.sh [tag [bench.py arguments]]   (default: tag "pmc", the default workload)
TAG=${1:-pmc}; shift || true
OUT=$REPO/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export LORA_BENCH_CACHE=${LORA_BENCH_CACHE:-/dev/shm/lora_bench}
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-grad-line --min-seconds 0 $*"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_FLAT SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -- $CMD > "$OUT/g$i.log" 2>&1
  echo "group $i rc=$? : $grp"
done
