#!/bin/bash
# round 5, call 7c: MSDA forms after the shared branch-free geometry (LDS-staged form pipelined too): tests + timing
O=gpurun_out/r7c; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "msda" > $O/k.log 2>&1; rc=$?; echo "kernel rc=$rc"; tail -3 $O/k.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/msda_bench.py > $O/msda_bench.txt 2>$O/err.txt; echo "bench rc=$?"; cat $O/msda_bench.txt
