#!/bin/bash
mkdir -p gpurun_out/r5k
S="4600x384x1536 7680x384x1536 3300x384x1536 4600x256x2048 2400x768x3072"
ALL_TILES=1 timeout -k 10 300 python tools/gemm_shape_bench.py $S > gpurun_out/r5k/shapes.txt 2>&1; echo rc=$?; grep -v amdgpu gpurun_out/r5k/shapes.txt | cut -c1-200
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > gpurun_out/r5k/k.log 2>&1; echo "gemm tests rc=$?"; tail -2 gpurun_out/r5k/k.log
B="--no-cpu-baseline --no-roofline --no-variants --steps 80"
for i in 1 2 3; do timeout -k 10 200 python bench.py $B > gpurun_out/r5k/cfg2_$i.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5k/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
PY
