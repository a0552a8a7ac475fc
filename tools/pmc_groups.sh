#!/bin/bash
# rocprofv3 --pmc passes over a one-stream bench run, one pass per quoted counter group. Usage: tools/pmc_groups.sh <outdir> "<bench args>" "GROUP 1" "GROUP 2" ...
OUT=$1; ARGS=$2; shift; shift
export TMPDIR=/tmp
mkdir -p $OUT
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pass$i -- python bench.py --cpu-samples 0 --steps 1 --warmup 0 $ARGS > $OUT/pass$i.json 2> $OUT/pass$i.err || echo "pass $i ($grp) failed: $(grep -i "error\|invalid\|not" $OUT/pass$i.err | head -2)"
done
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
