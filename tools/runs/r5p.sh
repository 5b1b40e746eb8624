#!/bin/bash
mkdir -p gpurun_out/r5p
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "groupnorm or gelu" > gpurun_out/r5p/k.log 2>&1; echo "k rc=$?"; tail -2 gpurun_out/r5p/k.log
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "small_matches_reference or config2_fullsize" > gpurun_out/r5p/e.log 2>&1; echo "e rc=$?"; tail -2 gpurun_out/r5p/e.log
