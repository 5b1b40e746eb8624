#!/bin/bash
mkdir -p gpurun_out/r5g
O=gpurun_out/r5g
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "window_attention or swin_attn" > $O/ktests.log 2>&1
rc=$?; echo "attn tests rc=$rc"; tail -5 $O/ktests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python tools/window_attn_bench.py > $O/attn2d.txt 2>&1; echo "bench rc=$?"; grep -v amdgpu $O/attn2d.txt
B="--no-cpu-baseline --no-roofline --no-variants"
C5="--backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --steps 20"
for i in 1 2; do
TCE_LIB=tools/runs/libtce_prev.so timeout -k 10 300 python bench.py $B $C5 > $O/cfg5_prev_$i.json 2>/dev/null
timeout -k 10 300 python bench.py $B $C5 > $O/cfg5_new_$i.json 2>/dev/null
TCE_LIB=tools/runs/libtce_prev.so timeout -k 10 200 python bench.py $B --steps 80 > $O/cfg2_prev_$i.json 2>/dev/null
timeout -k 10 200 python bench.py $B --steps 80 > $O/cfg2_new_$i.json 2>/dev/null
done
TCE_LIB=tools/runs/libtce_prev.so timeout -k 10 300 python bench.py $B $C5 --arith-policy cfg5_mixed > $O/cfg5m_prev.json 2>/dev/null
timeout -k 10 300 python bench.py $B $C5 --arith-policy cfg5_mixed > $O/cfg5m_new.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5g/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "reference or oracle" > $O/e2e.log 2>&1
rc=$?; echo "e2e rc=$rc"; tail -3 $O/e2e.log
