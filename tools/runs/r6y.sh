#!/bin/bash
# round 5, call y: MSDA gather with several sampling points in flight: tests (bit-identity with the loop), timing
O=gpurun_out/r6y; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "msda" > $O/k.log 2>&1; rc=$?; echo "kernel rc=$rc"; tail -3 $O/k.log
timeout -k 10 200 python tools/msda_bench.py > $O/msda_bench.txt 2>$O/err.txt; echo "bench rc=$?"; cat $O/msda_bench.txt
