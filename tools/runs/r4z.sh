#!/bin/bash
mkdir -p gpurun_out/r4z
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "window_attention" > gpurun_out/r4z/ktests.log 2>&1
rc=$?; echo "attn tests rc=$rc"; tail -5 gpurun_out/r4z/ktests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python tools/window_attn3d_bench.py > gpurun_out/r4z/attn3d_new.txt 2>&1; echo "bench rc=$?"; cat gpurun_out/r4z/attn3d_new.txt
TCE_LIB=tools/runs/libtce_prev.so timeout -k 10 300 python tools/window_attn3d_bench.py > gpurun_out/r4z/attn3d_prev.txt 2>&1; echo "bench prev rc=$?"; cat gpurun_out/r4z/attn3d_prev.txt
timeout -k 10 400 python -m pytest tests/test_e2e_gpu.py -x -q -k "video_swin" > gpurun_out/r4z/e2e.log 2>&1
rc=$?; echo "e2e video rc=$rc"; tail -3 gpurun_out/r4z/e2e.log
