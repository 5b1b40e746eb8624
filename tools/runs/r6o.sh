#!/bin/bash
# round 5, call o: contrastive kernel test (eps clamp per norm, like ATen) + the vis/contrastive e2e fixture
O=gpurun_out/r6o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -x -q -m gpu -k "contrastive" > $O/t.log 2>&1; echo "rc=$?"; tail -4 $O/t.log
