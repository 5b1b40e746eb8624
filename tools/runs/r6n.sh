#!/bin/bash
# round 5, call n: replay soak on owned executables (800 replays at config 2, every output bit-identical to the eager pass)
O=gpurun_out/r6n; mkdir -p $O
timeout -k 10 600 python tools/replay_soak.py > $O/replay_soak.txt 2>&1; echo "soak rc=$?"; tail -4 $O/replay_soak.txt
