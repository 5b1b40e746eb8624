#!/bin/bash
# round 4: kernel traces + bench lines on the current build
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4x
mkdir -p $O
guard() { rc=$1; if [ $rc -ne 0 ]; then echo "step failed (rc=$rc): stopping"; exit $rc; fi; }
B="--no-cpu-baseline --no-roofline --no-variants"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof2 -o cfg2 -- python3 bench.py --steps 60 $B > $O/bench_cfg2_under_rocprof.json 2> $O/p2.err
rc=$?; echo "prof cfg2 rc=$rc"; guard $rc
python tools/rocpd_stats.py $O/prof2/cfg2_results.db 66 > $O/kernel_stats_cfg2.csv
python tools/latency_summary.py $O/prof2/cfg2_results.db "BASELINE config 2" $O/latency_bound.json
python tools/overlap_stats.py $O/prof2/cfg2_results.db > $O/overlap_cfg2.txt 2>&1
python tools/timeline.py $O/prof2/cfg2_results.db 3 0 > $O/timeline_cfg2.txt 2>&1
rm -rf $O/prof2
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o cfg1 -- python3 bench.py --backbone resnet50 --frames 1 --steps 60 $B > $O/bench_cfg1_under_rocprof.json 2> $O/p1.err
rc=$?; echo "prof cfg1 rc=$rc"; guard $rc
python tools/rocpd_stats.py $O/prof1/cfg1_results.db 66 > $O/kernel_stats_cfg1.csv
python tools/latency_summary.py $O/prof1/cfg1_results.db "BASELINE config 1" $O/latency_bound.json
rm -rf $O/prof1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof3 -o cfg3 -- python3 bench.py --backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --steps 30 $B > $O/bench_cfg3_under_rocprof.json 2> $O/p3.err
rc=$?; echo "prof cfg3 rc=$rc"; guard $rc
python tools/rocpd_stats.py $O/prof3/cfg3_results.db 36 > $O/kernel_stats_cfg3.csv
rm -rf $O/prof3
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof5 -o cfg5 -- python3 bench.py --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --steps 20 $B > $O/bench_cfg5_under_rocprof.json 2> $O/p5.err
rc=$?; echo "prof cfg5 rc=$rc"; guard $rc
python tools/rocpd_stats.py $O/prof5/cfg5_results.db 26 > $O/kernel_stats_cfg5.csv
rm -rf $O/prof5
timeout -k 10 400 python bench.py --backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/c3.err; rc=$?; echo "cfg3 rc=$rc"; guard $rc
timeout -k 10 400 python bench.py --backbone resnet50 --frames 1 --no-cpu-baseline > $O/bench_cfg1.json 2> $O/c1.err; rc=$?; echo "cfg1 rc=$rc"; guard $rc
timeout -k 10 400 python bench.py --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --no-cpu-baseline --no-variants > $O/bench_cfg5_f16x3.json 2> $O/c5.err; rc=$?; echo "cfg5 rc=$rc"; guard $rc
timeout -k 10 400 python bench.py --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --arith-policy cfg5_mixed --no-cpu-baseline --no-variants > $O/bench_cfg5_mixed.json 2> $O/c5m.err; rc=$?; echo "cfg5 mixed rc=$rc"; guard $rc
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4x/bench_cfg*.json')):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f, b['value'], b['ms_per_step'], {k:v for k,v in b.items() if k.startswith('value_')})
    except Exception as e: print(f, 'ERR', e)
print(open('gpurun_out/r4x/latency_bound.json').read())
PY
exit 0
