#!/bin/bash
mkdir -p gpurun_out/r4y
timeout -k 10 300 python tools/replay_soak.py --reps 400 > gpurun_out/r4y/soak_g1.txt 2>&1; rc=$?; echo "soak g1 rc=$rc"; tail -2 gpurun_out/r4y/soak_g1.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 300 python tools/replay_soak.py --reps 150 --group 4 > gpurun_out/r4y/soak_g4.txt 2>&1; rc=$?; echo "soak g4 rc=$rc"; tail -2 gpurun_out/r4y/soak_g4.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 300 python tools/replay_soak.py --reps 100 --backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --group 2 > gpurun_out/r4y/soak_v_g2.txt 2>&1; rc=$?; echo "soak video g2 rc=$rc"; tail -2 gpurun_out/r4y/soak_v_g2.txt
