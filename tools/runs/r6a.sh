#!/bin/bash
# round 5, call a: every GEMM launch of a config-2 clip (shape, stage, us) + config 5, and the baseline bench
O=gpurun_out/r6a; mkdir -p $O
timeout -k 10 200 python tools/gemm_shapes.py --max-rows 1000000 > $O/gemm_shapes_cfg2.txt 2> $O/e1.txt; echo "rc=$?"
timeout -k 10 200 python tools/gemm_shapes.py --max-rows 1000000 --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 > $O/gemm_shapes_cfg5.txt 2> $O/e2.txt; echo "rc=$?"
timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2> $O/e3.txt; echo "rc=$?"
tail -c 600 $O/bench.json
