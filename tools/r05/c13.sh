#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c13; mkdir -p $O
(time timeout 700 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^  File" | tail -25) > $O/pytest.txt 2>&1
timeout 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -30 $O/pytest.txt | cut -c1-250; python -c "import json;d=json.load(open('$O/bench.json'));print(d['value'],d['roofline']['frac'],d['config']['bit_exact_vs_expected'])"
