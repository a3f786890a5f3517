#!/bin/bash
# round 3, second GPU call: the whole GPU suite (no -x), the SF11 / SF12 gradient-vs-oracle diagnostic, same-box A/B of the walker3
# variants (old = round 2's geometry and one-window acquisition rounds; t512only; new = default; new15 = SF11 at 512 threads too; k8)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c2; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
(time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -150) > $O/pytest.log 2>&1
timeout 900 python tools/r03_diag_sf11.py > $O/diag.txt 2>&1
for sf in 9 10 11 12; do
  echo "## sf$sf" >> $O/ab.txt
  REPS=1 tools/ab.sh "--config 3 --sf $sf --steps 16 --warmup 3" ab/old.so ab/t512only.so ab/new.so ab/new15.so ab/k8.so >> $O/ab.txt 2>&1
done
for sf in 9 11 12; do
LORA_HIP_DEBUG=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --config 3 --sf $sf --depth 1 2> $O/dbg_sf$sf.log >/dev/null
done
tail -3 $O/pytest.log; cat $O/diag.txt; cat $O/ab.txt
