#!/bin/bash
# round 5, call 7m: C <= 128 fused MLP as half (128-row) workgroups, two per CU: kernel tests, timing, e2e, A/B
O=gpurun_out/r7m; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ffn" > $O/k.log 2>&1; rc=$?; echo "kernel rc=$rc"; tail -3 $O/k.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/ffn_bench.py --shapes 72000x96x384:gelu:in,122880x96x384:gelu:in,256800x128x512:gelu:in,36000x96x384:gelu:in > $O/ffn_bench.txt 2>$O/err.txt; echo "bench rc=$?"; cat $O/ffn_bench.txt
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "matches_reference or race_free or taps or replay or video" > $O/e2e.log 2>&1; rc=$?; echo "e2e rc=$rc"; tail -3 $O/e2e.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-variants > $O/b1.json 2>>$O/err.txt || exit 1
python -c "import json;d=json.load(open('$O/b1.json'));print('cfg2 B=1', d['value'], d['ms_per_step'])"
