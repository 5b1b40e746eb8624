#!/bin/bash
mkdir -p gpurun_out/r5l
S="4600x384x1536 1200x2304x768 1200x768x768 1200x3072x768 24100x256x256 4600x2048x256 1100x256x256 18000x192x384 4600x384x768 1200x256x256 18000x256x192"
for i in 1 2; do
timeout -k 10 300 python tools/gemm_shape_bench.py $S > gpurun_out/r5l/base_$i.txt 2>&1
TCE_LIB=tools/runs/libtce_alt.so timeout -k 10 300 python tools/gemm_shape_bench.py $S > gpurun_out/r5l/alt_$i.txt 2>&1
done
for f in gpurun_out/r5l/*.txt; do echo == $f; grep -v amdgpu $f | cut -c1-100; done
B="--no-cpu-baseline --no-roofline --no-variants --steps 60"
for i in 1 2; do
timeout -k 10 200 python bench.py $B > gpurun_out/r5l/cfg2_base_$i.json 2>/dev/null
TCE_LIB=tools/runs/libtce_alt.so timeout -k 10 200 python bench.py $B > gpurun_out/r5l/cfg2_alt_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5l/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
PY
