#!/bin/bash
# MIPT_WORK_RUN sweep (run of consecutive samples of a pixel handed out together). Usage: tools/sweep_work_run.sh [bench args]
for rep in 1 2; do for r in 64 256; do MIPT_WORK_RUN=$r timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-samples 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_time_s']
print('run $r', d['value'], k['generate'], k['extend'], k['shade'], k['shadow'], k['mis'])"; done; done
