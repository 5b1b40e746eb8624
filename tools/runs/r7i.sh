#!/bin/bash
# round 5, call 7i: kernel trace of config 3 (Video-Swin-T, 8 x 384 x 640) and config 1 (ResNet-50, 1 frame): per-kernel table + alone time
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r7i; mkdir -p $O
B="--no-cpu-baseline --no-roofline --no-variants"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/p3 -o c3 -- python3 bench.py --backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --steps 40 $B > $O/bench_cfg3_under_rocprof.json 2> $O/p3.err
rc=$?; echo "prof cfg3 rc=$rc"; [ $rc -eq 0 ] || exit 1
python tools/rocpd_stats.py $O/p3/c3_results.db 46 > $O/kernel_stats_cfg3.csv
python tools/alone_time.py $O/p3/c3_results.db 30 > $O/alone_cfg3.txt
rm -rf $O/p3
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/p1 -o c1 -- python3 bench.py --backbone resnet50 --frames 1 --steps 60 $B > $O/bench_cfg1_under_rocprof.json 2> $O/p1.err
rc=$?; echo "prof cfg1 rc=$rc"; [ $rc -eq 0 ] || exit 1
python tools/rocpd_stats.py $O/p1/c1_results.db 66 > $O/kernel_stats_cfg1.csv
python tools/alone_time.py $O/p1/c1_results.db 40 > $O/alone_cfg1.txt
rm -rf $O/p1
head -32 $O/alone_cfg3.txt
