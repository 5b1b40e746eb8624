#!/bin/bash
mkdir -p gpurun_out/r5n
for i in 1 2; do
timeout -k 10 300 python tools/rowlin384_bench.py 512 16200 > gpurun_out/r5n/a_$i.txt 2>&1; grep "^M=" gpurun_out/r5n/a_$i.txt
timeout -k 10 300 python tools/rowlin384_bench.py 384 18000 > gpurun_out/r5n/b_$i.txt 2>&1; grep "^M=" gpurun_out/r5n/b_$i.txt
done
