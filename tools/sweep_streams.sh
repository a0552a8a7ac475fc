#!/bin/bash
# Sweep sub-renderer count and total pool size on the default 1024-spp frame.
for k in 2 3 4 6; do for p in 8388608 16777216 33554432; do
MIPT_STREAMS=$k timeout -k 10 120 python bench.py --steps 1 --cpu-samples 0 --exclusive-spp 0 --pool $p > gpurun_out/sw.json 2>gpurun_out/sw.err && python -c "
import json;d=json.load(open('gpurun_out/sw.json'));print('k',$k,'pool',$p,d['value'],d['film_mean_per_sample'],d['roofline']['launches'],d['seconds'])" >> gpurun_out/sweep.txt
done; done
cat gpurun_out/sweep.txt
