#!/bin/bash
# round 5, call 7a: MSDA gather with four sampling points in flight: bit-identity of the forms, timing, e2e, A/B of the clip
O=gpurun_out/r7a; mkdir -p $O
timeout -k 10 200 python tools/scratch/msda_dbg.py > $O/dbg.txt 2>&1; tail -4 $O/dbg.txt
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "msda" > $O/k.log 2>&1; rc=$?; echo "kernel rc=$rc"; tail -3 $O/k.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/msda_bench.py > $O/msda_bench.txt 2>$O/err.txt; echo "bench rc=$?"; cat $O/msda_bench.txt
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "matches_reference or race_free or taps or replay or video or padded or group" > $O/e2e.log 2>&1; rc=$?; echo "e2e rc=$rc"; tail -3 $O/e2e.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-variants > $O/b1.json 2>>$O/err.txt || exit 1
python -c "import json;d=json.load(open('$O/b1.json'));print('cfg2 B=1', d['value'], d['ms_per_step'])"
