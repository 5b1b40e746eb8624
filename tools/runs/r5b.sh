#!/bin/bash
# compile-time arithmetic mode in every matrix-core kernel: tests, then A/B against the previous build in one call
mkdir -p gpurun_out/r5b
O=gpurun_out/r5b
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q > $O/ktests.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -4 $O/ktests.log
if [ $rc -ne 0 ]; then exit 1; fi
B="--no-cpu-baseline --no-roofline --no-variants"
for i in 1 2; do
  TCE_LIB=tools/runs/libtce_head.so timeout -k 10 200 python bench.py $B --steps 80 > $O/cfg2_head_$i.json 2>/dev/null
  timeout -k 10 200 python bench.py $B --steps 80 > $O/cfg2_new_$i.json 2>/dev/null
done
TCE_LIB=tools/runs/libtce_head.so timeout -k 10 200 python bench.py $B --steps 40 --group 4 > $O/cfg2g4_head.json 2>/dev/null
timeout -k 10 200 python bench.py $B --steps 40 --group 4 > $O/cfg2g4_new.json 2>/dev/null
C3="--backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --steps 30"
TCE_LIB=tools/runs/libtce_head.so timeout -k 10 300 python bench.py $B $C3 > $O/cfg3_head.json 2>/dev/null
timeout -k 10 300 python bench.py $B $C3 > $O/cfg3_new.json 2>/dev/null
C5="--backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --steps 20"
TCE_LIB=tools/runs/libtce_head.so timeout -k 10 300 python bench.py $B $C5 > $O/cfg5_head.json 2>/dev/null
timeout -k 10 300 python bench.py $B $C5 > $O/cfg5_new.json 2>/dev/null
TCE_LIB=tools/runs/libtce_head.so timeout -k 10 300 python bench.py $B $C5 --arith-policy cfg5_mixed > $O/cfg5m_head.json 2>/dev/null
timeout -k 10 300 python bench.py $B $C5 --arith-policy cfg5_mixed > $O/cfg5m_new.json 2>/dev/null
C1="--backbone resnet50 --frames 1 --steps 80"
TCE_LIB=tools/runs/libtce_head.so timeout -k 10 300 python bench.py $B $C1 > $O/cfg1_head.json 2>/dev/null
timeout -k 10 300 python bench.py $B $C1 > $O/cfg1_new.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5b/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py -x -q > $O/e2e.log 2>&1
rc=$?; echo "e2e rc=$rc"; tail -4 $O/e2e.log
