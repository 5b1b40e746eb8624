#!/bin/bash
# round 5, call 7s: Swin stage output norms applied in the branches that consume them: e2e (fixtures, races, taps, groups), A/B
O=gpurun_out/r7s; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu > $O/e2e.log 2>&1; rc=$?; echo "e2e rc=$rc"; tail -3 $O/e2e.log
[ $rc -eq 0 ] || exit 1
B="--no-cpu-baseline --no-roofline --no-variants"
for rep in 1 2 3; do for c in 1 0; do
  TCE_DEFER_OUT_NORM=$c timeout -k 10 200 python bench.py --steps 200 --warmup 20 $B > $O/c2_defer${c}_$rep.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/c2_defer${c}_$rep.json'));print('cfg2 defer_out_norm=$c', d['value'], d['ms_per_step'])"
done; done
