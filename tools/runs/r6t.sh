#!/bin/bash
# round 5, call t: fused FFN with the hidden extent split (p blocks -> p + 1 workgroups): kernel test, timing, e2e, A/B
O=gpurun_out/r6t; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ffn" > $O/k.log 2>&1; rc=$?; echo "kernel rc=$rc"; tail -5 $O/k.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/ffn_bench.py > $O/ffn_bench.txt 2>$O/err.txt; echo "bench rc=$?"; tail -12 $O/ffn_bench.txt
timeout -k 10 500 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "matches_reference or race_free or taps or replay" > $O/e2e.log 2>&1; rc=$?; echo "e2e rc=$rc"; tail -5 $O/e2e.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do for c in 1 0; do
  TCE_FFN_SPLIT=$c timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-variants > $O/b1_split${c}_$rep.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/b1_split${c}_$rep.json'));print('B=1 split=$c', d['value'], d['ms_per_step'])"
done; done
