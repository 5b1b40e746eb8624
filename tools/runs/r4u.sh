#!/bin/bash
mkdir -p gpurun_out/r4u
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "rowlin" > gpurun_out/r4u/ktests.log 2>&1
rc=$?; echo "rowlin tests rc=$rc"; tail -5 gpurun_out/r4u/ktests.log
if [ $rc -ne 0 ]; then exit 1; fi
C5="--backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --no-cpu-baseline --no-variants --no-roofline --steps 30 --warmup 5"
for i in 1 2; do
for k in 1 0; do
  TCE_ROWLIN_K512=$k timeout -k 10 300 python bench.py $C5 > gpurun_out/r4u/cfg5_k512_${k}_$i.json 2> gpurun_out/r4u/err.txt
  rc=$?; echo "cfg5 k512=$k rc=$rc"; if [ $rc -ne 0 ]; then tail -10 gpurun_out/r4u/err.txt; exit 1; fi
done; done
TCE_ROWLIN_K512=1 timeout -k 10 300 python bench.py $C5 --arith-policy cfg5_mixed > gpurun_out/r4u/cfg5_mixed_k512_1.json 2> gpurun_out/r4u/err.txt
TCE_ROWLIN_K512=0 timeout -k 10 300 python bench.py $C5 --arith-policy cfg5_mixed > gpurun_out/r4u/cfg5_mixed_k512_0.json 2> gpurun_out/r4u/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4u/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
