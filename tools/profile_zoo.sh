#!/bin/bash
# Kernel-trace profiles of the two "every material" workloads: the textured zoo and the (untextured) material zoo with
# Uber / Disney / glass / metal / substrate / translucent / mix. Usage: tools/profile_zoo.sh <tag>
T=$1
export TMPDIR=/tmp
python tools/make_textured_scene.py /tmp/texzoo > /dev/null
python - <<PY
import sys; sys.path.insert(0, "tests")
import scenes_text as st
open("/tmp/matzoo.pbrt", "w").write(st.material_zoo(res=700, spp=256, depth=6))
PY
for w in texzoo matzoo; do
  if [ $w == texzoo ]; then SC=/tmp/texzoo/textured-zoo.pbrt; else SC=/tmp/matzoo.pbrt; fi
  OUT=gpurun_out/${T}_$w; mkdir -p $OUT
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-samples 0 --scene $SC --spp 256 > $OUT/bench.json 2> $OUT/bench.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 1 --warmup 1 --cpu-samples 0 --scene $SC --spp 256 > $OUT/trace_bench.json 2> $OUT/trace.err
  find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
  find $OUT -name "*kernel_trace.csv" -size +20M -delete
  python - <<PY
import json, csv
d = json.load(open("$OUT/bench.json")); print("$w", d["value"], "Mray/s", d["ms_per_step"], "ms", d["roofline"]["kernel_time_s"])
for r in list(csv.DictReader(open("$OUT/kernel_stats.csv")))[:9]:
    print("   %5.1f%%  %8.3f ms avg  x%s  %s" % (float(r["Percentage"]), float(r["AverageNs"]) / 1e6, r["Calls"], r["Name"].replace("(anonymous namespace)::", "")[:70]))
PY
done
