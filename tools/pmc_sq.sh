#!/bin/bash
# SQ / TCP counter passes of a one-stream bench run. Usage: tools/pmc_sq.sh <outdir> [bench args...]
OUT=$1; shift
ARGS=${@:-"--spp 256 --pool 8388608"}
export TMPDIR=/tmp
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pass$i -- python bench.py --cpu-samples 0 --steps 1 --warmup 0 $ARGS > $OUT/pass$i.json 2> $OUT/pass$i.err || echo "pass $i failed"
done
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
cat $OUT/pmc_summary.txt
