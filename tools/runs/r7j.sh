#!/bin/bash
# round 5, call 7j: bilinear resize + add + LayerNorm as one pass: kernel test, e2e, A/B at B=1 and G=8
O=gpurun_out/r7j; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "resize or layernorm" > $O/k.log 2>&1; rc=$?; echo "kernel rc=$rc"; tail -3 $O/k.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "matches_reference or race_free or taps or replay or video or padded or group" > $O/e2e.log 2>&1; rc=$?; echo "e2e rc=$rc"; tail -3 $O/e2e.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do for c in 1 0; do
  TCE_RESIZE_LN_FUSE=$c timeout -k 10 200 python bench.py --steps 40 --warmup 5 --group 8 --no-cpu-baseline --no-roofline --no-variants > $O/g8_fuse${c}_$rep.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/g8_fuse${c}_$rep.json'));print('G=8 resize_ln_fuse=$c', d['value'], d['ms_per_step'])"
done; done
for c in 1 0; do
  TCE_RESIZE_LN_FUSE=$c timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-variants > $O/b1_fuse${c}.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/b1_fuse${c}.json'));print('B=1 resize_ln_fuse=$c', d['value'], d['ms_per_step'])"
done
