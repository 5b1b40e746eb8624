#!/bin/bash
mkdir -p gpurun_out/r5i
timeout -k 10 300 python tools/gemm_shapes.py --max-rows 100000 > gpurun_out/r5i/gemm_shapes.txt 2>&1; echo rc=$?; grep -v amdgpu gpurun_out/r5i/gemm_shapes.txt | cut -c1-150
