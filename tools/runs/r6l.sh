#!/bin/bash
# round 5, call l: the N > 1 bench flow with its variants (value_group8 on every rank), both rehearsal forms, RCCL at world 1
O=gpurun_out/r6l; mkdir -p $O
timeout -k 10 1200 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "rehearsal or rccl or config4" > $O/t.log 2>&1; rc=$?; echo "rc=$rc"; tail -15 $O/t.log
