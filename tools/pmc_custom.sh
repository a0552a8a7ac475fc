#!/bin/bash
# One rocprofv3 --pmc pass with the given counters over a short one-stream bench.
# Usage: tools/pmc_custom.sh <outdir> COUNTER...
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p $OUT
MIPT_STREAMS=1 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pass1 -- python bench.py --cpu-samples 0 --steps 1 --warmup 0 --spp 64 --exclusive-spp 0 --pool 2097152 > $OUT/pass1.json 2> $OUT/pass1.err || echo "pass failed"
python tools/pmc_summary.py $OUT
