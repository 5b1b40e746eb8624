#!/bin/bash
# round 5, call 7y: the whole GPU suite on the current build + the default bench line
O=gpurun_out/r7y; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -8 $O/gpu_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench.json
