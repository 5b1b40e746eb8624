#!/bin/bash
mkdir -p gpurun_out/r5a
S="24100x256x2048:relu:out,96400x256x2048:relu:out,72000x96x384:gelu:in,18000x192x768:gelu:in"
for i in 1 2; do
timeout -k 10 200 python tools/ffn_bench.py --shapes $S --iters 30 > gpurun_out/r5a/ffn_base_$i.txt 2>&1; echo "base rc=$?"
TCE_LIB=tools/runs/libtce_alt.so timeout -k 10 200 python tools/ffn_bench.py --shapes $S --iters 30 > gpurun_out/r5a/ffn_alt_$i.txt 2>&1; echo "alt rc=$?"
done
for f in gpurun_out/r5a/ffn_*.txt; do echo == $f; grep -v amdgpu $f; done
B="--no-cpu-baseline --no-roofline --no-variants --steps 60"
for i in 1 2; do
timeout -k 10 200 python bench.py $B > gpurun_out/r5a/b_base_$i.json 2>/dev/null
TCE_LIB=tools/runs/libtce_alt.so timeout -k 10 200 python bench.py $B > gpurun_out/r5a/b_alt_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5a/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
