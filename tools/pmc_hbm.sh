#!/bin/bash
# HBM traffic counters (two rocprofv3 --pmc passes) of a one-stream bench run.
# Usage: tools/pmc_hbm.sh <outdir> [bench args...]   (default: 256 spp, 8M-slot pool)
OUT=$1; shift
ARGS=${@:-"--spp 256 --pool 8388608"}
export TMPDIR=/tmp
mkdir -p $OUT
i=0
for grp in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pass$i -- python bench.py --cpu-samples 0 --steps 1 --warmup 0 $ARGS > $OUT/pass$i.json 2> $OUT/pass$i.err || echo "pass $i failed"
done
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
cat $OUT/pmc_summary.txt
