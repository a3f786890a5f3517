#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c7; mkdir -p $O
for sf in 9 10 11 12; do LORA_HIP_W3_STAMPS=1 timeout 300 python tools/demod_bench.py $sf > $O/stamps_sf$sf.txt 2>&1; done
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -22 $O/stamps_sf9.txt; tail -12 $O/stamps_sf11.txt; tail -3 $O/smoke.txt
