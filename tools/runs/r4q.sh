#!/bin/bash
mkdir -p gpurun_out/r4q
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r4q/prof -o g8 -- python3 bench.py --steps 20 --warmup 6 --group 8 --no-cpu-baseline --no-variants --no-roofline > gpurun_out/r4q/bench_g8.json 2> gpurun_out/r4q/bench_g8.err
rc=$?; echo "prof rc=$rc"
if [ $rc -ne 0 ]; then tail -20 gpurun_out/r4q/bench_g8.err; exit 1; fi
python tools/rocpd_stats.py gpurun_out/r4q/prof/g8_results.db 208 > gpurun_out/r4q/kernel_stats_g8.csv
python tools/timeline.py gpurun_out/r4q/prof/g8_results.db 3 0 > gpurun_out/r4q/timeline_g8.txt 2>&1
rm -rf gpurun_out/r4q/prof
head -45 gpurun_out/r4q/kernel_stats_g8.csv | cut -c1-150
