#!/bin/bash
# round 5, call c: sample-then-project few-query MSDA (kernel test, e2e subset, A/B at B=1 and G=8), few-row threshold A/B
O=gpurun_out/r6c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "msda" > $O/k.log 2>&1; echo "kernel rc=$?"; tail -3 $O/k.log
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "matches_reference or race_free or flag_comb" > $O/e2e.log 2>&1; echo "e2e rc=$?"; tail -3 $O/e2e.log
B="python bench.py --steps 150 --no-cpu-baseline --no-roofline --no-variants"
for raw in 1 0 1 0; do TCE_MSDA_RAW=$raw timeout -k 10 200 $B > $O/b1_raw$raw.json 2>> $O/err.txt; python -c "import json;d=json.loads(open('$O/b1_raw$raw.json').read().strip().splitlines()[-1]);print('B=1 raw=$raw',d['value'],d['ms_per_step'])"; done
for cfg in "1 128" "0 256" "1 256" "1 64"; do set -- $cfg; TCE_MSDA_RAW=$1 TCE_FEWROW_MAX_ROWS=$2 timeout -k 10 300 $B --steps 40 --group 8 > $O/g8_$1_$2.json 2>> $O/err.txt; python -c "import json;d=json.loads(open('$O/g8_$1_$2.json').read().strip().splitlines()[-1]);print('G=8 raw=$1 fewrow<=$2',d['value'],d['ms_per_step'])"; done
