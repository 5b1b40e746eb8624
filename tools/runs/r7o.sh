#!/bin/bash
# round 5, call 7o: mixed 3x3 convolution launches (full rounds of 256-pixel workgroups + the remainder as 128-pixel ones): tests, timing, A/B
O=gpurun_out/r7o; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv3x3" > $O/k.log 2>&1; rc=$?; echo "kernel rc=$rc"; tail -3 $O/k.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/conv3_bench.py > $O/conv3_bench.txt 2>$O/err.txt; echo "bench rc=$?"; tail -12 $O/conv3_bench.txt
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "matches_reference or race_free or replay" > $O/e2e.log 2>&1; rc=$?; echo "e2e rc=$rc"; tail -3 $O/e2e.log
[ $rc -eq 0 ] || exit 1
B="--no-cpu-baseline --no-roofline --no-variants"
for rep in 1 2; do for c in 0 4; do
  TCE_CONV3_FORM=$c timeout -k 10 200 python bench.py --steps 200 --warmup 20 $B > $O/c2_form${c}_$rep.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/c2_form${c}_$rep.json'));print('cfg2 conv form=$c', d['value'], d['ms_per_step'])"
done; done
