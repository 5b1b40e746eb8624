#!/bin/bash
mkdir -p gpurun_out/r4r
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "clip_group or expressions_of_one" > gpurun_out/r4r/group_tests.log 2>&1
rc=$?; echo "group tests rc=$rc"; tail -25 gpurun_out/r4r/group_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
run() { name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-variants "$@" > gpurun_out/r4r/$name.json 2> gpurun_out/r4r/$name.err
  rc=$?; echo "$name rc=$rc"
  if [ $rc -ne 0 ]; then tail -15 gpurun_out/r4r/$name.err; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
run cfg2_g1 --steps 40 --warmup 8
run cfg2_g2 --steps 40 --warmup 8 --group 2
run cfg2_g2_same --steps 40 --warmup 8 --group 2 --same-clip
run cfg2_g4 --steps 40 --warmup 8 --group 4
run cfg2_g4_same --steps 40 --warmup 8 --group 4 --same-clip
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4r/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["config"].get("clips_per_forward"), d["config"].get("same_clip_in_group"))
    except Exception as e: print(f, "ERR", e)
PY
