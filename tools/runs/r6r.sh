#!/bin/bash
# round 5, call r: every tile forced at the 8-clip group's GEMM shapes (is the automatic choice right there?)
O=gpurun_out/r6r; mkdir -p $O
ALL_TILES=1 timeout -k 10 400 python tools/gemm_shape_bench.py 36800x384x384 192800x256x256 36800x1536x384 36800x1152x384 36800x384x1536 144000x192x384 144000x256x192 576000x256x96 9600x768x768 8800x256x256 9600x256x256 256x2304x768 256x3072x768 256x768x3072 256x768x768 36800x384x768 9600x256x2048 9600x2048x256 > $O/tiles.txt 2>$O/err.txt; echo "rc=$?"
cat $O/tiles.txt
