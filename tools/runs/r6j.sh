#!/bin/bash
# round 5, call j: whole GPU suite on the build with both convolution forms + the config-5 / group A/B of the automatic choice
O=gpurun_out/r6j; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -4 $O/gpu_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
B="--no-cpu-baseline --no-roofline --no-variants"
for w in 4 0 4 0; do TCE_CONV3_WAVES=$w timeout -k 10 300 python bench.py --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --steps 40 $B > $O/c5_w$w.json 2>> $O/err.txt; python -c "import json;d=json.loads(open('$O/c5_w$w.json').read().strip().splitlines()[-1]);print('cfg5 conv waves=$w (0 = automatic)',d['value'],d['ms_per_step'])"; done
for w in 4 0 4 0; do TCE_CONV3_WAVES=$w timeout -k 10 300 python bench.py --steps 30 --group 8 $B > $O/g8_w$w.json 2>> $O/err.txt; python -c "import json;d=json.loads(open('$O/g8_w$w.json').read().strip().splitlines()[-1]);print('cfg2 G=8 conv waves=$w',d['value'],d['ms_per_step'])"; done
