#!/bin/bash
# round 5, call 7t: kernel trace of the config-2 clip on the current build: alone time / idle-before per kernel + one clip's timeline
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r7t; mkdir -p $O
B="--no-cpu-baseline --no-roofline --no-variants"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof2 -o cfg2 -- python3 bench.py --steps 60 $B > $O/bench_cfg2_under_rocprof.json 2> $O/p2.err
rc=$?; echo "prof cfg2 rc=$rc"; [ $rc -eq 0 ] || exit 1
python tools/alone_time.py $O/prof2/cfg2_results.db 40 > $O/alone_cfg2.txt
python tools/timeline.py $O/prof2/cfg2_results.db 3 0 > $O/timeline_cfg2.txt
rm -rf $O/prof2
sort -k5 -n -r $O/alone_cfg2.txt | head -5
