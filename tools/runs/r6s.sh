#!/bin/bash
# round 5, call s: new tile rules (256x128 from 1.5 rounds at N >= 256) -- kernel tests, the one B=1 shape they move, A/B of the clip and the group
O=gpurun_out/r6s; mkdir -p $O
ALL_TILES=1 timeout -k 10 200 python tools/gemm_shape_bench.py 72000x256x96 72000x256x256 85570x256x256 102720x256x128 > $O/tiles.txt 2>$O/err.txt; echo "rc=$?"; cat $O/tiles.txt
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > $O/k.log 2>&1; rc=$?; echo "kernel rc=$rc"; tail -3 $O/k.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do for r in r5 r4; do
  TCE_GEMM_TILE_RULES=$r timeout -k 10 200 python bench.py --steps 40 --warmup 5 --group 8 --no-cpu-baseline --no-roofline --no-variants > $O/g8_${r}_$rep.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/g8_${r}_$rep.json'));print('G=8 rules=$r', d['value'], d['ms_per_step'])"
done; done
for r in r5 r4; do
  TCE_GEMM_TILE_RULES=$r timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-variants > $O/b1_$r.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/b1_$r.json'));print('B=1 rules=$r', d['value'], d['ms_per_step'])"
done
