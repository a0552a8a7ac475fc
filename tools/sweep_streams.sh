#!/bin/bash
# Sweep sub-renderer count and total pool size on the default 1024-spp frame.
# Usage: tools/sweep_streams.sh "<k list>" "<pool list (slots)>"
KS=${1:-"1 2 4"}; PS=${2:-"16777216 33554432 67108864"}
rm -f gpurun_out/sweep.txt
for k in $KS; do for p in $PS; do
MIPT_STREAMS=$k timeout -k 10 120 python bench.py --steps 2 --cpu-samples 0 --exclusive-spp 0 --pool $p > gpurun_out/sw.json 2>gpurun_out/sw.err && python -c "
import json;d=json.load(open('gpurun_out/sw.json'));r=d['roofline'];print('k',$k,'pool',$p>>20,'M',d['value'],'launches',r['launches'],'s',d['seconds'],'frac',r['frac'],'avg_ms',r['avg_launch_ms'])" >> gpurun_out/sweep.txt
done; done
cat gpurun_out/sweep.txt
