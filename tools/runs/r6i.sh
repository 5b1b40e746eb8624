#!/bin/bash
# round 5, call i: 3x3 convolution with 256-pixel (8-wave) workgroups, A/B against the 128-pixel form
O=gpurun_out/r6i; mkdir -p $O
TCE_CONV3_WAVES=8 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv3x3" > $O/k8.log 2>&1; rc=$?; echo "kernel tests (8 waves) rc=$rc"; tail -3 $O/k8.log
if [ $rc -ne 0 ]; then exit $rc; fi
for w in 4 8 4 8; do TCE_CONV3_WAVES=$w timeout -k 10 200 python tools/conv3_bench.py > $O/conv_w$w.txt 2>&1; echo "waves=$w rc=$?"; grep "px" $O/conv_w$w.txt | cut -c1-110; done
B="python bench.py --steps 150 --no-cpu-baseline --no-roofline --no-variants"
for w in 4 8 4 8; do TCE_CONV3_WAVES=$w timeout -k 10 200 $B > $O/b_w$w.json 2>> $O/err.txt; python -c "import json;d=json.loads(open('$O/b_w$w.json').read().strip().splitlines()[-1]);print('conv waves=$w',d['value'],d['ms_per_step'])"; done
