#!/bin/bash
O=gpurun_out/r6z; mkdir -p $O
timeout -k 10 200 python tools/scratch/msda_dbg.py > $O/dbg.txt 2>&1; cat $O/dbg.txt | tail -30
timeout -k 10 200 python tools/msda_bench.py > $O/msda_bench.txt 2>$O/err.txt; echo "bench rc=$?"; cat $O/msda_bench.txt
