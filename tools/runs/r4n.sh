#!/bin/bash
# clip groups: tests + A/B at config 2
mkdir -p gpurun_out/r4n
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "clip_group" > gpurun_out/r4n/group_tests.log 2>&1
rc=$?; echo "group tests rc=$rc"; tail -25 gpurun_out/r4n/group_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
for g in 1 2 3 4 1 2 4; do
  timeout -k 10 240 python bench.py --steps 60 --warmup 10 --group $g --no-cpu-baseline --no-variants > gpurun_out/r4n/bench_g${g}_$RANDOM.json 2> gpurun_out/r4n/bench.err
  rc=$?; echo "bench g=$g rc=$rc"
  if [ $rc -ne 0 ]; then tail -20 gpurun_out/r4n/bench.err; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4n/bench_g*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["config"].get("clips_per_forward"))
    except Exception as e: print(f, "ERR", e)
PY
