#!/bin/bash
# round 5, call g: exact-fp32 tiled GEMM after LDS-read pipelining + four waves per SIMD; kernel tests in f32; exact-mode clip
O=gpurun_out/r6g; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv or gelu" > $O/k.log 2>&1; echo "kernel rc=$?"; tail -3 $O/k.log
BENCH_GEMM_MODE=f32 timeout -k 10 400 python tools/gemm_shape_bench.py 24100x2048x256 24100x256x2048 72000x2048x256 24100x256x256 4600x1536x384 4600x384x1536 72000x256x96 18000x256x2304 1200x3072x768 > $O/gemm_f32.txt 2>&1; echo "rc=$?"; cat $O/gemm_f32.txt
timeout -k 10 300 python bench.py --gemm-mode f32 --steps 60 --no-cpu-baseline --no-roofline --no-variants > $O/bench_f32.json 2> $O/err.txt; python -c "import json;d=json.loads(open('$O/bench_f32.json').read().strip().splitlines()[-1]);print('exact f32',d['value'],d['ms_per_step'])"
