#!/bin/bash
mkdir -p gpurun_out/r4t
timeout -k 10 300 python tools/rowlin384_bench.py 512 > gpurun_out/r4t/rowlin512.txt 2>&1
echo rc=$?; cat gpurun_out/r4t/rowlin512.txt
