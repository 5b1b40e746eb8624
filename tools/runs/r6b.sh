#!/bin/bash
# round 5, call b: new tests (per-op reference fixtures, valid_indices, config-4 rehearsal) + whole GPU suite
O=gpurun_out/r6b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_perop_gpu.py -x -q -m gpu > $O/perop.log 2>&1; echo "perop rc=$?"; tail -5 $O/perop.log
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "valid_indices or config4 or clip_group" > $O/e2e_new.log 2>&1; echo "e2e_new rc=$?"; tail -5 $O/e2e_new.log
