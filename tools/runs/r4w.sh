#!/bin/bash
mkdir -p gpurun_out/r4w
timeout -k 10 300 python tools/fewrow_bench.py rows > gpurun_out/r4w/fewrow_rows.txt 2>&1
echo rc=$?; cat gpurun_out/r4w/fewrow_rows.txt
timeout -k 10 400 python -m pytest tests/test_e2e_gpu.py -x -q -k "two_rank_gloo" > gpurun_out/r4w/gloo.log 2>&1
echo "gloo rc=$?"; tail -3 gpurun_out/r4w/gloo.log
