#!/bin/bash
# round 5, call 7k: replay soak on the final build (split FFN counters, fused merges): config 3 (8 split launches per clip) and config 2
O=gpurun_out/r7k; mkdir -p $O
timeout -k 10 500 python tools/replay_soak.py --backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --reps 1500 > $O/soak_cfg3.txt 2>&1; echo "cfg3 rc=$?"; tail -3 $O/soak_cfg3.txt
timeout -k 10 400 python tools/replay_soak.py --reps 2000 > $O/soak_cfg2.txt 2>&1; echo "cfg2 rc=$?"; tail -3 $O/soak_cfg2.txt
timeout -k 10 300 python tools/replay_soak.py --reps 200 --group 8 > $O/soak_cfg2_g8.txt 2>&1; echo "g8 rc=$?"; tail -3 $O/soak_cfg2_g8.txt
