#!/bin/bash
# round 4, GPU call 12: one clip's kernel timeline; config 5 bench lines
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l
mkdir -p $O
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed (rc=$rc): stopping"; exit $rc; fi; }
B="--no-cpu-baseline --no-roofline --no-variants"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof2 -o cfg2 -- python3 bench.py --steps 40 $B > $O/bench_cfg2_under_rocprof.json 2> $O/p2.err
rc=$?; echo "prof cfg2 rc=$rc"; guard $rc
python tools/timeline.py $O/prof2/cfg2_results.db 3 0 > $O/timeline_cfg2.txt 2>&1
python tools/timeline.py $O/prof2/cfg2_results.db 5 0 > $O/timeline_cfg2_b.txt 2>&1
rm -rf $O/prof2
head -3 $O/timeline_cfg2.txt
timeout -k 10 400 python bench.py --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --no-cpu-baseline --no-variants > $O/bench_cfg5_f16x3.json 2> $O/c5.err; rc=$?; echo "cfg5 rc=$rc"; guard $rc
timeout -k 10 400 python bench.py --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --arith-policy cfg5_mixed --no-cpu-baseline --no-variants > $O/bench_cfg5_mixed.json 2> $O/c5m.err; rc=$?; echo "cfg5 mixed rc=$rc"; guard $rc
TCE_SWIN_FUSED=0 TCE_ROWLIN_K384=0 timeout -k 10 400 python bench.py --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --no-cpu-baseline --no-variants --no-roofline > $O/bench_cfg5_f16x3_off.json 2> $O/c5o.err; guard $?
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4l/bench_cfg5*.json')):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f, b['value'], b['ms_per_step'], (b.get('roofline') or {}).get('top_time_kernel'))
    except Exception as e: print(f, 'ERR', e)
PY
exit 0
