#!/bin/bash
# Path-pool size sweep on the killeroo frame. Usage: tools/sweep_pool.sh [slots ...]
for rep in 1 2; do for p in ${@:-67108864 100663296 134217728}; do timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-samples 0 --pool $p 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_time_s']
print('pool $p', d['value'], k['generate'], k['extend'], k['shade'], k['shadow'], k['mis'], d['config'].get('path_pool_gb'))"; done; done
