#!/bin/bash
# rocprofv3 PMC passes over a short bench run (one counter group per run, as the
# MI355X guide prescribes). Usage: tools/pmc_passes.sh <outdir> [bench args...]
set -e
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pass$i -- python bench.py --cpu-samples 0 "$@" > $OUT/pass$i.json 2> $OUT/pass$i.err || echo "pass $i failed"
done
python tools/pmc_summary.py $OUT
