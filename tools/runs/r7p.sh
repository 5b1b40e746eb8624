#!/bin/bash
O=gpurun_out/r7p; mkdir -p $O
timeout -k 10 200 python tools/conv3_bench.py > $O/conv3_bench.txt 2>$O/err.txt; echo "bench rc=$?"; grep "launcher" $O/conv3_bench.txt
