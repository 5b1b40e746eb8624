#!/bin/bash
# round 5, call k: LayerNorm prologue of the few-row kernel (frame-token path: norm1 / norm2 ride in the next projection launch)
O=gpurun_out/r6k; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "fewrow" > $O/k.log 2>&1; rc=$?; echo "kernel rc=$rc"; tail -3 $O/k.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py tests/test_perop_gpu.py -x -q -m gpu -k "matches_reference or race_free or flag_comb or taps or clip_group" > $O/e2e.log 2>&1; rc=$?; echo "e2e rc=$rc"; tail -3 $O/e2e.log
if [ $rc -ne 0 ]; then exit $rc; fi
B="python bench.py --steps 200 --no-cpu-baseline --no-roofline --no-variants"
for f in 1 0 1 0; do TCE_FTF_LN_FUSE=$f timeout -k 10 200 $B > $O/b_$f.json 2>> $O/err.txt; python -c "import json;d=json.loads(open('$O/b_$f.json').read().strip().splitlines()[-1]);print('ftf ln fuse=$f',d['value'],d['ms_per_step'])"; done
