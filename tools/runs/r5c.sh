#!/bin/bash
mkdir -p gpurun_out/r5c
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "window_attention" > gpurun_out/r5c/ktests.log 2>&1
rc=$?; echo "attn tests rc=$rc"; tail -3 gpurun_out/r5c/ktests.log
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2; do
TCE_LIB=tools/runs/libtce_prev.so timeout -k 10 300 python tools/window_attn3d_bench.py > gpurun_out/r5c/attn3d_prev_$i.txt 2>&1
timeout -k 10 300 python tools/window_attn3d_bench.py > gpurun_out/r5c/attn3d_new_$i.txt 2>&1
done
for f in gpurun_out/r5c/attn3d_*.txt; do echo == $f; grep "^T=" $f | cut -c1-100; done
