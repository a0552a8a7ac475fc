#!/bin/bash
# GPU parity tests, then short benches of the three BASELINE workloads. Usage: tools/quick_gpu.sh <tag> [pytest -k expr]
T=$1
K=${2:-""}
if [ -n "$K" ]; then timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/t_$T.log 2>&1; else timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_$T.log 2>&1; fi
tail -3 gpurun_out/t_$T.log
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-samples 0 > gpurun_out/qk_$T.json 2> gpurun_out/qk_$T.err
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-samples 0 --scene scenes/cornell-glass.pbrt --spp 4096 > gpurun_out/qc_$T.json 2> gpurun_out/qc_$T.err
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-samples 0 --procedural-tris 10000000 --spp 256 > gpurun_out/qp_$T.json 2> gpurun_out/qp_$T.err
python - <<PY
import json
for f in ("qk_$T","qc_$T","qp_$T"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); r=d["roofline"]
        print(f, d["value"], "Mray/s", d["ms_per_step"], "ms film", d["film_mean_per_sample"], r["kernel_time_s"], "frac", r["frac"], "nodes", r["nodes_per_ray"], r["tri_tests_per_ray"])
    except Exception as e: print(f, "failed", e)
PY
