#!/bin/bash
# Short benches of the two all-materials workloads (material zoo, textured zoo; 700x700, 256 spp). Usage: tools/bench_zoo.sh <tag>
T=$1
python -c "
import sys; sys.path.insert(0, 'tests')
import scenes_text as st
open('/tmp/matzoo.pbrt', 'w').write(st.material_zoo(res=700, spp=256, depth=6))"
python tools/make_textured_scene.py /tmp/texzoo > /dev/null
for w in matzoo texzoo; do
  if [ $w == texzoo ]; then SC=/tmp/texzoo/textured-zoo.pbrt; else SC=/tmp/matzoo.pbrt; fi
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-samples 0 --scene $SC --spp 256 > gpurun_out/zoo_${T}_$w.json 2> gpurun_out/zoo_${T}_$w.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/zoo_${T}_$w.json")); r = d["roofline"]
    print("$w", d["value"], "Mray/s", d["ms_per_step"], "ms", r["kernel_time_s"], "shade frac", r["classes"]["shade (k_shade), 0.96 KB/vertex"]["frac"], "mean", d["film_mean_per_sample"])
except Exception as e: print("$w failed", e)
PY
done
