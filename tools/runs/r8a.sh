#!/bin/bash
# round 5, call 8a: the round's A/B switches all OFF at once: the alternative paths still pass the reference fixtures
O=gpurun_out/r8a; mkdir -p $O
TCE_FFN_SPLIT=0 TCE_GN_UP_FUSE=0 TCE_DEFER_OUT_NORM=0 TCE_RESIZE_LN_FUSE=0 TCE_MSDA_RAW=0 timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "matches_reference or race_free or replay" > $O/e2e_off.log 2>&1; rc=$?; echo "switches off rc=$rc"; tail -3 $O/e2e_off.log
TCE_XATTN_FFN_CHAIN=1 TCE_FTF_LN_FUSE=1 timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "matches_reference or race_free or replay" > $O/e2e_on.log 2>&1; rc=$?; echo "off-by-default switches on rc=$rc"; tail -3 $O/e2e_on.log
