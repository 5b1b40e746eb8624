#!/bin/bash
# clip groups at configs 3 / 5 / 1, and groups x clips in flight at config 2
mkdir -p gpurun_out/r4o
run() { # name, args...
  name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-variants "$@" > gpurun_out/r4o/$name.json 2> gpurun_out/r4o/$name.err
  rc=$?; echo "$name rc=$rc"
  if [ $rc -ne 0 ]; then tail -15 gpurun_out/r4o/$name.err; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
run cfg2_g4_c1 --steps 40 --warmup 8 --group 4
run cfg2_g4_c2 --steps 40 --warmup 8 --group 4 --clips-in-flight 2
run cfg2_g2_c2 --steps 40 --warmup 8 --group 2 --clips-in-flight 2
run cfg2_g1_c2 --steps 40 --warmup 8 --group 1 --clips-in-flight 2
run cfg3_g1 --backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --steps 30 --warmup 6 --group 1
run cfg3_g2 --backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --steps 30 --warmup 6 --group 2
run cfg3_g4 --backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --steps 20 --warmup 6 --group 4
run cfg5_g1 --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --steps 20 --warmup 5 --group 1
run cfg5_g2 --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --steps 20 --warmup 5 --group 2
run cfg1_g1 --backbone resnet50 --frames 1 --steps 60 --warmup 10 --group 1
run cfg1_g4 --backbone resnet50 --frames 1 --steps 60 --warmup 10 --group 4
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4o/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["config"].get("clips_per_forward"), d["config"].get("clips_in_flight_per_gpu"))
    except Exception as e: print(f, "ERR", e)
PY
