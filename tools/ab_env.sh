#!/bin/bash
# Same-box A/B of environment settings on the default bench. Usage: tools/ab_env.sh <tag> "VAR=a" "VAR=b" ... [-- bench args]
T=$1; shift
ARGS=""
SETS=()
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; ARGS="$@"; break; fi
  SETS+=("$1"); shift
done
for rep in 1 2; do
  for st in "${SETS[@]}"; do
    env $st timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-samples 0 $ARGS > gpurun_out/ab_${T}.json 2> gpurun_out/ab_${T}.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab_${T}.json")); r=d["roofline"]
    print("$st", d["value"], "Mray/s", d["ms_per_step"], "ms", r["kernel_time_s"])
except Exception as e: print("$st", "failed", e)
PY
  done
done
