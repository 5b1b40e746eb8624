#!/bin/bash
# round 4, GPU call 1: hazard checker on the real program, RCCL at world 1, ranks per GPU, AQL packets of one replay
set -o pipefail
mkdir -p gpurun_out/r4a
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed (rc=$rc): stopping"; exit $rc; fi; }
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -s -k "race_free or dropped_join or rccl_world1" > gpurun_out/r4a/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a gpurun_out/r4a/tests.log; tail -5 gpurun_out/r4a/tests.log; guard $rc
timeout -k 10 600 python tools/ranks_per_gpu.py --ranks 1 2 3 --steps 250 > gpurun_out/r4a/ranks_per_gpu.txt 2>&1
rc=$?; echo "ranks rc=$rc"; tail -5 gpurun_out/r4a/ranks_per_gpu.txt; guard $rc
timeout -k 10 300 python tools/aql_log.py gpurun_out/r4a > gpurun_out/r4a/aql_tool.log 2>&1
rc=$?; echo "aql rc=$rc"; tail -3 gpurun_out/r4a/aql_tool.log; guard $rc
exit 0
