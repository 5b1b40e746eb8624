#!/bin/bash
mkdir -p gpurun_out/r5j
S="4600x384x1536 1200x2304x768 1200x768x3072 1200x768x768 24100x256x256 4600x2048x256 1100x256x256 18000x192x384"
ALL_TILES=1 timeout -k 10 300 python tools/gemm_shape_bench.py $S > gpurun_out/r5j/f16x3.txt 2>&1; echo rc=$?
BENCH_GEMM_MODE=f16 ALL_TILES=1 timeout -k 10 300 python tools/gemm_shape_bench.py $S > gpurun_out/r5j/f16.txt 2>&1; echo rc=$?
for f in gpurun_out/r5j/*.txt; do echo == $f; grep -v amdgpu $f | cut -c1-200; done
