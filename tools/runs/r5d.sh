#!/bin/bash
mkdir -p gpurun_out/r5d
S="24100x256x2048:relu:out,96400x256x2048:relu:out,72000x96x384:gelu:in,18000x192x768:gelu:in"
for i in 1 2; do
timeout -k 10 200 python tools/ffn_bench.py --shapes $S --iters 30 > gpurun_out/r5d/ffn_base_$i.txt 2>&1; echo "base rc=$?"
TCE_LIB=tools/runs/libtce_alt.so timeout -k 10 200 python tools/ffn_bench.py --shapes $S --iters 30 > gpurun_out/r5d/ffn_alt_$i.txt 2>&1; echo "alt rc=$?"
done
for f in gpurun_out/r5d/ffn_*.txt; do echo == $f; grep -v amdgpu $f | cut -c1-110; done
