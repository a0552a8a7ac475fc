#!/bin/bash
# GPU parity tests, then one-stream and default short benches. Usage: tools/quick_gpu.sh <tag>
T=$1
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/t_$T.log 2>&1; tail -3 gpurun_out/t_$T.log
MIPT_STREAMS=1 timeout -k 10 200 python bench.py --steps 1 --warmup 1 --spp 256 --pool 2097152 --cpu-samples 0 --exclusive-spp 0 > gpurun_out/q1_$T.json 2> gpurun_out/q1_$T.err
timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-samples 0 --exclusive-spp 0 > gpurun_out/q4_$T.json 2> gpurun_out/q4_$T.err
python - <<PY
import json
for f in ("q1_$T","q4_$T"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); r=d["roofline"]
        print(f, d["value"], d["film_mean_per_sample"], r["kernel_time_s"], "frac", r["frac"], "nodes", r["nodes_per_ray"], r["tri_tests_per_ray"])
    except Exception as e: print(f, "failed", e)
PY
