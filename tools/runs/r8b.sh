#!/bin/bash
# round 5, call 8b: kernel table of the exact-fp32 pass (bench.py --gemm-mode f32): where do its 14.5 ms go?
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r8b; mkdir -p $O
B="--no-cpu-baseline --no-roofline --no-variants"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/p -o f32 -- python3 bench.py --gemm-mode f32 --steps 30 $B > $O/bench_f32_under_rocprof.json 2> $O/p.err
rc=$?; echo "prof rc=$rc"; [ $rc -eq 0 ] || exit 1
python tools/rocpd_stats.py $O/p/f32_results.db 36 > $O/kernel_stats_f32.csv
python tools/alone_time.py $O/p/f32_results.db 20 > $O/alone_f32.txt
rm -rf $O/p
head -24 $O/alone_f32.txt
