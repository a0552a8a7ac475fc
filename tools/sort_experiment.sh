#!/bin/bash
# Experiment: traversal kernel durations with the ray queues sorted by (origin cell, direction octant) before the launch.
# Needs build_variants/lib_sort.so (tools/build_variants.sh "sort:-DMIPT_SORT_EXPERIMENT"). Usage: tools/sort_experiment.sh
export TMPDIR=/tmp
export MIPT_HIP_LIB=$PWD/build_variants/lib_sort.so
for wl in "killeroo:--spp 256" "procedural:--procedural-tris 10000000 --spp 64"; do
  name=${wl%%:*}; args=${wl#*:}
  for m in 0 2 4 1 7; do
    OUT=gpurun_out/sortexp_${name}_$m; rm -rf $OUT; mkdir -p $OUT
    MIPT_SORT=$m timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 1 --warmup 1 --cpu-samples 0 $args > $OUT/bench.json 2> $OUT/err.log
    f=$(find $OUT -name "*kernel_stats.csv" | head -1)
    python - <<PY
import csv, json
rows = {r["Name"]: r for r in csv.DictReader(open("$f"))}
def avg(pat):
    for n, r in rows.items():
        if pat in n: return float(r["AverageNs"]) / 1e6
    return float("nan")
d = json.load(open("$OUT/bench.json"))
print("$name MIPT_SORT=$m  k_trav<0> %.3f ms  k_trav<1> %.3f ms  k_trav<2> %.3f ms  k_shade(diffuse) %.3f ms  film mean %.6f" % (avg("k_trav<0"), avg("k_trav<1"), avg("k_trav<2"), avg("k_shade<2, 3859"), d["film_mean_per_sample"]))
PY
    find $OUT -name "*kernel_trace.csv" -delete
  done
done
