#!/bin/bash
mkdir -p gpurun_out/r5e
timeout -k 10 300 python tools/ffn_ablate.py > gpurun_out/r5e/ffn_ablate.txt 2>&1; echo rc=$?; grep -v amdgpu gpurun_out/r5e/ffn_ablate.txt | cut -c1-160
timeout -k 10 300 python tools/ffn_ablate.py --M 96400 > gpurun_out/r5e/ffn_ablate_96400.txt 2>&1; echo rc=$?; grep -v amdgpu gpurun_out/r5e/ffn_ablate_96400.txt | cut -c1-160
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "rowlin or ffn or conv3x3 or xattn or patch" > gpurun_out/r5e/ktests.log 2>&1; echo "ktests rc=$?"; tail -3 gpurun_out/r5e/ktests.log
