#!/bin/bash
# Same-box A/B of one environment switch: tools/ab_env.sh VAR  (runs the short benches with VAR unset, then VAR=1).
V=$1
for mode in off on; do
  if [ $mode = on ]; then export $V=1; fi
  MIPT_STREAMS=1 timeout -k 10 200 python bench.py --steps 1 --warmup 1 --spp 256 --pool 8388608 --cpu-samples 0 --exclusive-spp 0 > gpurun_out/v1.json 2> gpurun_out/v1.err
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-samples 0 --exclusive-spp 0 > gpurun_out/v4.json 2> gpurun_out/v4.err
  python - <<PY
import json
out=["$V $mode"]
for f in ("v1","v4"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); r=d["roofline"]["kernel_time_s"]
        out.append("%s %.1f (g %.3f t0 %.3f e %.3f sh %.3f s %.3f m %.3f) mean %.6f"%(f,d["value"],r["generate"],r["trav0"],r["extend"],r["shade"],r["shadow"],r["mis"],d["film_mean_per_sample"]))
    except Exception as e: out.append(f+" -")
print(" | ".join(out))
PY
done
