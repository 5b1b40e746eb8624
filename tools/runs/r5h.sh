#!/bin/bash
mkdir -p gpurun_out/r5h
O=gpurun_out/r5h
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q > $O/ktests.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -5 $O/ktests.log
if [ $rc -ne 0 ]; then exit 1; fi
S="72000x96x384:gelu:in,18000x192x768:gelu:in,256800x128x512:gelu:in,64200x256x1024:gelu:in"
for i in 1 2; do
TCE_LIB=tools/runs/libtce_prev.so timeout -k 10 200 python tools/ffn_bench.py --shapes $S --iters 30 > $O/ffn_prev_$i.txt 2>&1
timeout -k 10 200 python tools/ffn_bench.py --shapes $S --iters 30 > $O/ffn_new_$i.txt 2>&1
done
for f in $O/ffn_*.txt; do echo == $f; grep -v amdgpu $f | cut -c1-125; done
B="--no-cpu-baseline --no-roofline --no-variants"
C5="--backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --steps 20"
C3="--backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --steps 30"
for i in 1 2; do
TCE_LIB=tools/runs/libtce_prev.so timeout -k 10 200 python bench.py $B --steps 80 > $O/cfg2_prev_$i.json 2>/dev/null
timeout -k 10 200 python bench.py $B --steps 80 > $O/cfg2_new_$i.json 2>/dev/null
done
TCE_LIB=tools/runs/libtce_prev.so timeout -k 10 300 python bench.py $B $C5 > $O/cfg5_prev.json 2>/dev/null
timeout -k 10 300 python bench.py $B $C5 > $O/cfg5_new.json 2>/dev/null
TCE_LIB=tools/runs/libtce_prev.so timeout -k 10 300 python bench.py $B $C3 > $O/cfg3_prev.json 2>/dev/null
timeout -k 10 300 python bench.py $B $C3 > $O/cfg3_new.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5h/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
