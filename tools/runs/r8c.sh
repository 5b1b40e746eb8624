#!/bin/bash
# round 5, call 8c: 600 distinct clip shapes through a 3-entry graph cache on the final build (replay = eager bit for bit at every shape:
# the split FFN's plans, the mixed convolution launches, the half-workgroup MLP at arbitrary row counts)
O=gpurun_out/r8c; mkdir -p $O
timeout -k 10 900 python tools/graph_cycle.py --shapes 600 > $O/graph_cycle.txt 2>&1; echo "rc=$?"; tail -6 $O/graph_cycle.txt
