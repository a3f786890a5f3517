#!/bin/bash
# tools/flip_all.sh - the round's decision-flip sweep (tools/decision_flip_sweep.py) on one box: the marginal band of the reference's gates at decimation 8 (SF7-SF11, both
# estimators) + high SNR at SF9-SF12 (where the closed-form fine_sync of SF9 and up is active) + decimation 4 / 2 on walker2's LD builds.  -> gpurun_out/flips.jsonl
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; : > gpurun_out/flips.jsonl
S="timeout 900 python tools/decision_flip_sweep.py"
for d in 2 0; do
  $S --sf 7,8 --snr 30:38:1 --packets 96 --demod $d >> gpurun_out/flips.jsonl 2>> gpurun_out/flips.err
  $S --sf 9,10 --snr 30:38:1 --packets 48 --demod $d >> gpurun_out/flips.jsonl 2>> gpurun_out/flips.err
  $S --sf 11 --snr 30:38:2 --packets 24 --demod $d >> gpurun_out/flips.jsonl 2>> gpurun_out/flips.err
  $S --sf 9,10,11 --snr 45:60:15 --packets 32 --demod $d >> gpurun_out/flips.jsonl 2>> gpurun_out/flips.err
  $S --sf 12 --snr 45:60:15 --packets 12 --demod $d >> gpurun_out/flips.jsonl 2>> gpurun_out/flips.err
  for r in 5e5 2.5e5; do
    $S --sf 7,8,9 --snr 34:50:2 --packets 64 --demod $d --samp-rate $r >> gpurun_out/flips.jsonl 2>> gpurun_out/flips.err
  done
done
wc -l gpurun_out/flips.jsonl; python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/flips.jsonl")]
print("packets", sum(r["packets"] for r in rows), "oracle frames", sum(r["oracle_frames"] for r in rows), "trace diffs", sum(r["streams_with_trace_diff"] for r in rows), "frame diffs", sum(r["frames_differ"] for r in rows))
for r in rows:
    if r["streams_with_trace_diff"] or r["frames_differ"]: print(r)
PY
