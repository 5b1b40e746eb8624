#!/bin/bash
mkdir -p gpurun_out/r5f
S="16200x1536x512 16200x512x512 16200x2048x512 16200x512x2048 4050x3072x1024 4050x1024x4096 64800x384x128 72000x256x256 24100x512x256"
for i in 1 2; do
timeout -k 10 200 python tools/gemm_shape_bench.py $S > gpurun_out/r5f/base_$i.txt 2>&1; echo "base rc=$?"
TCE_LIB=tools/runs/libtce_alt.so timeout -k 10 200 python tools/gemm_shape_bench.py $S > gpurun_out/r5f/alt_$i.txt 2>&1; echo "alt rc=$?"
done
for f in gpurun_out/r5f/*.txt; do echo == $f; grep -v amdgpu $f | cut -c1-120; done
B="--no-cpu-baseline --no-roofline --no-variants --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --steps 20"
for i in 1 2; do
timeout -k 10 300 python bench.py $B > gpurun_out/r5f/cfg5_base_$i.json 2>/dev/null
TCE_LIB=tools/runs/libtce_alt.so timeout -k 10 300 python bench.py $B > gpurun_out/r5f/cfg5_alt_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5f/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
