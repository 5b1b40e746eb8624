#!/bin/bash
# round 4, GPU call 2: full GPU suite on the round's changes, text-branch early-edge A/B, config-1 / config-2 kernel traces
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b
mkdir -p $O
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed (rc=$rc): stopping"; exit $rc; fi; }
B="--no-cpu-baseline --no-roofline --no-variants"
for i in 1 2; do
  TCE_TEXT_EARLY_EDGE=0 timeout -k 10 200 python bench.py $B > $O/ab_edge0_$i.json 2> $O/ab.err; guard $?
  TCE_TEXT_EARLY_EDGE=1 timeout -k 10 200 python bench.py $B > $O/ab_edge1_$i.json 2> $O/ab.err; guard $?
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4b/ab_edge*.json')):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f, b['value'], b['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log; guard $rc
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o cfg1 -- python3 bench.py --backbone resnet50 --frames 1 --steps 60 $B > $O/bench_cfg1_under_rocprof.json 2> $O/p1.err
rc=$?; echo "prof cfg1 rc=$rc"; guard $rc
python tools/rocpd_stats.py $O/prof1/cfg1_results.db 66 > $O/kernel_stats_cfg1.csv
python tools/latency_summary.py $O/prof1/cfg1_results.db "BASELINE config 1" $O/latency_bound.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof2 -o cfg2 -- python3 bench.py --steps 60 $B > $O/bench_cfg2_under_rocprof.json 2> $O/p2.err
rc=$?; echo "prof cfg2 rc=$rc"; guard $rc
python tools/rocpd_stats.py $O/prof2/cfg2_results.db 66 > $O/kernel_stats_cfg2.csv
python tools/latency_summary.py $O/prof2/cfg2_results.db "BASELINE config 2" $O/latency_bound.json
rm -rf $O/prof1 $O/prof2
exit 0
