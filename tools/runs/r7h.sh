#!/bin/bash
# round 5, call 7h: GroupNorm apply fused with the top-down merge (up-sample + add): kernel test, e2e, A/B
O=gpurun_out/r7h; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "groupnorm or resize or ffn_fused_split" > $O/k.log 2>&1; rc=$?; echo "kernel rc=$rc"; tail -3 $O/k.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "matches_reference or race_free or taps or replay or video or padded or group" > $O/e2e.log 2>&1; rc=$?; echo "e2e rc=$rc"; tail -3 $O/e2e.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do for c in 1 0; do
  TCE_GN_UP_FUSE=$c timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-variants > $O/b1_fuse${c}_$rep.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/b1_fuse${c}_$rep.json'));print('cfg2 B=1 gn_up_fuse=$c', d['value'], d['ms_per_step'])"
done; done
