#!/bin/bash
mkdir -p gpurun_out/r5t
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py -x -q -k "clip_group or expressions or run_video or ragged" > gpurun_out/r5t/e.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r5t/e.log
