#!/bin/bash
mkdir -p gpurun_out/r4p
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "clip_group" > gpurun_out/r4p/group_tests.log 2>&1
rc=$?; echo "group tests rc=$rc"; tail -25 gpurun_out/r4p/group_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
run() { # name, args...
  name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-variants "$@" > gpurun_out/r4p/$name.json 2> gpurun_out/r4p/$name.err
  rc=$?; echo "$name rc=$rc"
  if [ $rc -ne 0 ]; then tail -15 gpurun_out/r4p/$name.err; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
run cfg2_g4 --steps 40 --warmup 8 --group 4
run cfg2_g6 --steps 30 --warmup 8 --group 6
run cfg2_g8 --steps 30 --warmup 8 --group 8
run cfg1_g4 --backbone resnet50 --frames 1 --steps 60 --warmup 10 --group 4
run cfg1_g8 --backbone resnet50 --frames 1 --steps 60 --warmup 10 --group 8
run cfg1_g16 --backbone resnet50 --frames 1 --steps 40 --warmup 10 --group 16
run cfg1_g32 --backbone resnet50 --frames 1 --steps 30 --warmup 10 --group 32
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4p/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["config"].get("clips_per_forward"), d["config"].get("clips_in_flight_per_gpu"))
    except Exception as e: print(f, "ERR", e)
PY
