#!/bin/bash
# round 3, first GPU call: the whole GPU suite on the new code (RCCL world 1, stream latency, gradient / implicit fast kernels,
# reference-made full-size fixtures), first bench lines of the gradient kernels beside the FFT ones, per-state clocks of SF11
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/c1; mkdir -p $O
export LORA_BENCH_CACHE=/dev/shm/lora_bench
(time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -60) > $O/pytest.log 2>&1
for args in "" "--demod 0" "--config 3 --sf 8 --packets 1024" "--config 3 --sf 8 --packets 1024 --demod 0" "--config 3 --sf 9" "--config 3 --sf 9 --demod 0" "--config 3 --sf 11" "--config 3 --sf 11 --demod 0"; do
  echo "## $args" >> $O/bench.jsonl
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $args 2>>$O/bench.err | tail -1 >> $O/bench.jsonl
done
for sf in 9 11 12; do
LORA_HIP_DEBUG=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --config 3 --sf $sf --depth 1 2> $O/dbg_sf$sf.log >/dev/null
done
tail -5 $O/pytest.log
# same-box A/B: 512-thread x 256-register walker3 (LORA_W3_T512_MASK=15) against the default build
for sf in 9 10 11 12; do
  echo "## sf$sf" >> $O/ab.txt
  REPS=1 tools/ab.sh "--config 3 --sf $sf --steps 16 --warmup 3" ab/base.so ab/t512.so >> $O/ab.txt 2>&1
done
cat $O/ab.txt
