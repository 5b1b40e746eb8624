#!/bin/bash
mkdir -p gpurun_out/r4v
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "run_video_expressions or ragged_groups or video_driver" > gpurun_out/r4v/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -25 gpurun_out/r4v/tests.log
