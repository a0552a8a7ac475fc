#!/bin/bash
# Same-box A/B of named build_variants libraries on the killeroo frame (two rounds). Usage: tools/ab_libs.sh name1 name2 ... [-- bench args]
NAMES=(); ARGS=""
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; ARGS="$@"; break; fi; NAMES+=("$1"); shift; done
for rep in 1 2; do
for n in "${NAMES[@]}"; do
  MIPT_HIP_LIB=$PWD/build_variants/lib_$n.so timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-samples 0 $ARGS > gpurun_out/ab.json 2> gpurun_out/ab.err
  grep "k_shade stamps" gpurun_out/ab.err | tail -1
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab.json")); r=d["roofline"]["kernel_time_s"]
    print("%-24s %.1f Mray/s (g %.3f t0 %.3f e %.3f sh %.3f s %.3f m %.3f) mean %.6f" % ("$n", d["value"], r["generate"], r["trav0"], r["extend"], r["shade"], r["shadow"], r["mis"], d["film_mean_per_sample"]))
except Exception as e: print("$n", "failed", e)
PY
done
done
