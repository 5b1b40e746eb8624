#!/bin/bash
mkdir -p gpurun_out/r5u
O=gpurun_out/r5u
timeout -k 10 300 python bench.py --steps 3000 --no-variants --no-cpu-baseline --no-roofline > $O/long_b1.json 2> $O/e1.txt; echo "rc=$?"
timeout -k 10 300 python bench.py --steps 400 --group 8 --no-variants --no-cpu-baseline --no-roofline > $O/long_g8.json 2> $O/e2.txt; echo "rc=$?"
timeout -k 10 300 python bench.py --steps 600 --group 4 --same-clip --no-variants --no-cpu-baseline --no-roofline > $O/long_g4same.json 2> $O/e3.txt; echo "rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5u/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["steps"], d["graphs"])
PY
rocm-smi --showmeminfo vram 2>/dev/null | head -5
