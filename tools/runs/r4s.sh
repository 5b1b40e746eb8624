#!/bin/bash
mkdir -p gpurun_out/r4s
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r4s/gpu_tests.log 2>&1
rc=$?; echo "gpu tests rc=$rc"; tail -30 gpurun_out/r4s/gpu_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4s/smoke.log 2>&1
rc=$?; echo "smoke rc=$rc"; tail -3 gpurun_out/r4s/smoke.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 500 python bench.py > gpurun_out/r4s/bench_n1.json 2> gpurun_out/r4s/bench.err
rc=$?; echo "bench rc=$rc"; tail -c 3000 gpurun_out/r4s/bench_n1.json
