#!/bin/bash
mkdir -p gpurun_out/r5s
O=gpurun_out/r5s
B="--no-cpu-baseline --no-roofline --no-variants --steps 80"
for i in 1 2; do
timeout -k 10 200 python bench.py $B > $O/base_$i.json 2>/dev/null
TCE_BENCH_STREAM_PRIORITY=-1 timeout -k 10 200 python bench.py $B > $O/replayhi_$i.json 2>$O/err1.txt
TCE_BENCH_STREAM_PRIORITY=-1 TCE_MAIN_PRIORITY=-1 timeout -k 10 200 python bench.py $B > $O/replayhi_mainhi_$i.json 2>$O/err2.txt
done
tail -3 $O/err1.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5s/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f,"ERR",e)
PY
