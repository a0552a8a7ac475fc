#!/bin/bash
# Where k_shade's time goes: one-translation-unit builds with the matte / plastic instances alone (MIPT_HOT_ONLY), each
# with one part of the kernel taken out (MIPT_EXP_*: the films are wrong, only the kernel-class times are read), then a
# same-box A/B on the killeroo frame. Usage: tools/shade_experiments.sh build | run
case "$1" in
build)
  rm -f build_variants/lib_*.so
  tools/build_variants.sh "x0_base:-DMIPT_HOT_ONLY" "x1_nospec:-DMIPT_HOT_ONLY -DMIPT_EXP_NOSPEC" "x2_nostore:-DMIPT_HOT_ONLY -DMIPT_EXP_NOSTORE" \
     "x3_nospec_nostore:-DMIPT_HOT_ONLY -DMIPT_EXP_NOSPEC -DMIPT_EXP_NOSTORE" "x4_flattri:-DMIPT_HOT_ONLY -DMIPT_EXP_FLATTRI" \
     "x5_fastrng:-DMIPT_HOT_ONLY -DMIPT_EXP_FASTRNG" "x6_floatlibm:-DMIPT_HOT_ONLY -DMIPT_EXP_FLOATLIBM" \
     "x7_all:-DMIPT_HOT_ONLY -DMIPT_EXP_NOSPEC -DMIPT_EXP_NOSTORE -DMIPT_EXP_FLATTRI -DMIPT_EXP_FASTRNG -DMIPT_EXP_FLOATLIBM" ;;
run)
  ONLY_K=1 tools/run_variants2.sh ;;
esac
