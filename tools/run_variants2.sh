#!/bin/bash
# Same-box A/B of every build_variants/lib_*.so on the killeroo (1024 spp) and procedural 10M (256 spp) workloads, twice.
for rep in 1 2; do
for f in build_variants/lib_*.so; do
  n=$(basename $f .so)
  MIPT_HIP_LIB=$PWD/$f timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-samples 0 > gpurun_out/v1.json 2> gpurun_out/v1.err
  if [ -z "$ONLY_K" ]; then MIPT_HIP_LIB=$PWD/$f timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-samples 0 --procedural-tris 10000000 --spp 256 > gpurun_out/v4.json 2> gpurun_out/v4.err; else rm -f gpurun_out/v4.json; fi
  python - <<PY
import json
out=["$n"]
for f in ("v1","v4"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); r=d["roofline"]["kernel_time_s"]
        out.append("%s %.1f (g %.3f t0 %.3f e %.3f sh %.3f s %.3f m %.3f) mean %.6f"%(f,d["value"],r["generate"],r["trav0"],r["extend"],r["shade"],r["shadow"],r["mis"],d["film_mean_per_sample"]))
    except Exception as e: out.append(f+" -")
print(" | ".join(out))
PY
done
done
