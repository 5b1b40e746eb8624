#!/bin/bash
mkdir -p gpurun_out/r5r
O=gpurun_out/r5r
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
B="--no-cpu-baseline --no-roofline --no-variants --steps 80"
for i in 1 2; do
timeout -k 10 200 python bench.py $B > $O/base_$i.json 2>/dev/null
TCE_MAIN_PRIORITY=-1 timeout -k 10 200 python bench.py $B > $O/mainhi_$i.json 2>$O/err1.txt
TCE_SIDE_PRIORITY=-1 timeout -k 10 200 python bench.py $B > $O/sidehi_$i.json 2>$O/err2.txt
TCE_MAIN_PRIORITY=-1 TCE_SIDE_PRIORITY=0 timeout -k 10 200 python bench.py $B > $O/mainhi_side0_$i.json 2>$O/err3.txt
done
tail -3 $O/err1.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5r/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f,"ERR",e)
PY
