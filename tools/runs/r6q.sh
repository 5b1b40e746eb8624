#!/bin/bash
# round 5, call q: every tiled-GEMM launch of an 8-clip forward (shape, tile, time) -- is the tile choice right at the group's shapes?
O=gpurun_out/r6q; mkdir -p $O
timeout -k 10 300 python tools/gemm_shapes.py --group 8 --max-rows 100000000 > $O/g8.txt 2>$O/err.txt; echo "rc=$?"
timeout -k 10 300 python tools/gemm_shapes.py --group 1 --max-rows 100000000 > $O/g1.txt 2>>$O/err.txt; echo "rc=$?"
tail -3 $O/g8.txt
