#!/bin/bash
# Sweep sub-renderer count and total pool size on the default 1024-spp frame (two repetitions each).
rm -f gpurun_out/sweep.txt
for rep in 1 2; do for k in 1 2 3 4; do for p in 16777216 33554432 50331648; do
MIPT_STREAMS=$k timeout -k 10 120 python bench.py --steps 2 --cpu-samples 0 --exclusive-spp 0 --pool $p > gpurun_out/sw.json 2>gpurun_out/sw.err && python -c "
import json;d=json.load(open('gpurun_out/sw.json'));print('rep',$rep,'k',$k,'pool',$p>>20,'M',d['value'],d['roofline']['launches'],d['seconds'])" >> gpurun_out/sweep.txt
done; done; done
cat gpurun_out/sweep.txt
