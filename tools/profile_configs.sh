#!/bin/bash
# Bench line + rocprofv3 kernel trace + HBM PMC passes for the three single-GPU BASELINE workloads
# (configs[1] killeroo 1024 spp, configs[2] Cornell glass 4096 spp, configs[3] stand-in: procedural 10M triangles 256 spp).
# "matzoo" / "texzoo": the all-materials workloads (Uber, Disney, glass, metal, substrate, translucent, mix / the image-textured
# zoo), 700x700, 256 spp -- the shading instances north_star names.
# Usage: tools/profile_configs.sh <tag> [workload ...]     -> gpurun_out/<tag>_<workload>/
TAG=$1; shift
WL=${@:-"killeroo cornell procedural"}
export TMPDIR=/tmp
for w in $WL; do
  case $w in
    killeroo)   ARGS="" ;;
    cornell)    ARGS="--scene scenes/cornell-glass.pbrt --spp 4096" ;;
    procedural) ARGS="--procedural-tris 10000000 --spp 256" ;;
    matzoo)     python -c "
import sys; sys.path.insert(0, 'tests')
import scenes_text as st
open('/tmp/matzoo.pbrt', 'w').write(st.material_zoo(res=700, spp=256, depth=6))"
                ARGS="--scene /tmp/matzoo.pbrt --spp 256" ;;
    texzoo)     python tools/make_textured_scene.py /tmp/texzoo > /dev/null
                ARGS="--scene /tmp/texzoo/textured-zoo.pbrt --spp 256" ;;
    *) echo "unknown workload $w"; exit 1 ;;
  esac
  OUT=gpurun_out/${TAG}_$w
  mkdir -p $OUT
  echo "== $w: bench"
  timeout -k 10 500 python bench.py --steps 3 --warmup 1 $ARGS > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
  tools/profile_round.sh $OUT $ARGS > $OUT/profile.log 2>&1 || { echo "profile failed"; tail -5 $OUT/profile.log; exit 1; }
  python - <<PY
import json
d = json.load(open("$OUT/bench.json")); r = d["roofline"]
print("$w", d["value"], "Mray/s", d["ms_per_step"], "ms", "frac", r["frac"], "nodes", r["nodes_per_ray"], "cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"])
print(r["kernel_time_s"])
PY
done
