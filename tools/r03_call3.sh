#!/bin/bash
# round 3, third GPU call: the whole GPU suite after the fixes; SQ counters of the SF11 / SF12 walkers
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c3; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
(time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -250) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
tools/pmc_walker.sh sq_sf12 --config 3 --sf 12 > $O/pmc12.log 2>&1
python tools/pmc_summary.py gpurun_out/sq_sf12 > $O/sq_sf12.json 2>/dev/null
tools/pmc_walker.sh sq_sf11 --config 3 --sf 11 > $O/pmc11.log 2>&1
python tools/pmc_summary.py gpurun_out/sq_sf11 > $O/sq_sf11.json 2>/dev/null
rm -rf gpurun_out/sq_sf12 gpurun_out/sq_sf11
for sf in 9 10 11 12; do python bench.py --config 3 --sf $sf --steps 16 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench.jsonl; done
