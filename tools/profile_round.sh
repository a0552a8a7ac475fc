#!/bin/bash
# rocprofv3 kernel-trace summary + HBM traffic counters of the default bench workload (one frame each).
# Usage: tools/profile_round.sh <outdir under gpurun_out> [bench args...]
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 1 --warmup 1 --cpu-samples 0 "$@" > $OUT/trace_bench.json 2> $OUT/trace.err || echo trace failed
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $OUT/pass1 -- python bench.py --steps 1 --warmup 0 --cpu-samples 0 "$@" > $OUT/pass1.json 2> $OUT/pass1.err || echo pass1 failed
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/pass2 -- python bench.py --steps 1 --warmup 0 --cpu-samples 0 "$@" > $OUT/pass2.json 2> $OUT/pass2.err || echo pass2 failed
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
cat $OUT/kernel_stats.csv; cat $OUT/pmc_summary.txt
