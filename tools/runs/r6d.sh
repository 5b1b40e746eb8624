#!/bin/bash
# round 5, call d: graph executables destroyed on eviction (600 shapes), 128x64 GEMM depth A/B, stage-3 fc2 split-K A/B,
# exact-fp32 GEMM efficiency at the heavy shapes
O=gpurun_out/r6d; mkdir -p $O
timeout -k 10 1200 python tools/graph_cycle.py --shapes 600 > $O/graph_cycle.txt 2> $O/graph_cycle.err; echo "graph_cycle rc=$?"; tail -3 $O/graph_cycle.txt; tail -5 $O/graph_cycle.err
S="4600x384x1536 4600x1536x384 4600x384x384 24100x256x256 18000x192x384 4600x384x768 1200x768x1536 4600x256x2048 4600x2048x256 14400x160x256"
TCE_GEMM_12864_DEPTH=1 timeout -k 10 300 python tools/gemm_shape_bench.py $S > $O/gemm_depth1.txt 2>&1; echo "rc=$?"
TCE_GEMM_12864_DEPTH=3 timeout -k 10 300 python tools/gemm_shape_bench.py $S > $O/gemm_depth3.txt 2>&1; echo "rc=$?"
paste -d'|' $O/gemm_depth1.txt $O/gemm_depth3.txt | cut -c1-230
BENCH_GEMM_MODE=f32 timeout -k 10 400 python tools/gemm_shape_bench.py 24100x2048x256 24100x256x2048 72000x2048x256 24100x256x256 4600x1536x384 4600x384x1536 72000x256x96 18000x256x2304 1200x3072x768 > $O/gemm_f32.txt 2>&1; echo "rc=$?"; cat $O/gemm_f32.txt
B="python bench.py --steps 150 --no-cpu-baseline --no-roofline --no-variants"
for cfg in "1 1" "3 1" "3 2" "1 1" "3 2"; do set -- $cfg; TCE_GEMM_12864_DEPTH=$1 TCE_SWIN3_FC2_SPLITK=$2 timeout -k 10 200 $B > $O/b_$1_$2.json 2>> $O/err.txt; python -c "import json;d=json.loads(open('$O/b_$1_$2.json').read().strip().splitlines()[-1]);print('depth=$1 fc2 splitk=$2',d['value'],d['ms_per_step'])"; done
