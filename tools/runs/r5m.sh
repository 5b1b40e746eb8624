#!/bin/bash
mkdir -p gpurun_out/r5m
timeout -k 10 300 python tools/rowlin384_bench.py 512 5600 16200 32400 > gpurun_out/r5m/rowlin512.txt 2>&1; echo rc=$?; grep "^M=" gpurun_out/r5m/rowlin512.txt
timeout -k 10 300 python tools/rowlin384_bench.py 384 4600 7680 18000 > gpurun_out/r5m/rowlin384.txt 2>&1; echo rc=$?; grep "^M=" gpurun_out/r5m/rowlin384.txt
