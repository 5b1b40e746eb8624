#!/bin/bash
# round 4, GPU call 10: one-launch xattn pack, then the round's bench lines and kernel traces
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j
mkdir -p $O
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed (rc=$rc): stopping"; exit $rc; fi; }
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "xattn" > $O/ktests.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -3 $O/ktests.log; guard $rc
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "swin_t_small or config2_fullsize or race_free or longer_than_32" > $O/e2e.log 2>&1
rc=$?; echo "e2e rc=$rc"; tail -3 $O/e2e.log; guard $rc
B="--no-cpu-baseline --no-roofline --no-variants"
for i in 1 2; do
  TCE_XATTN_PACK_FUSED=0 timeout -k 10 200 python bench.py $B > $O/ab_xp0_$i.json 2> $O/ab.err; guard $?
  TCE_XATTN_PACK_FUSED=1 timeout -k 10 200 python bench.py $B > $O/ab_xp1_$i.json 2> $O/ab.err; guard $?
done
timeout -k 10 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; rc=$?; echo "bench rc=$rc"; guard $rc
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof2 -o cfg2 -- python3 bench.py --steps 60 $B > $O/bench_cfg2_under_rocprof.json 2> $O/p2.err
rc=$?; echo "prof cfg2 rc=$rc"; guard $rc
python tools/rocpd_stats.py $O/prof2/cfg2_results.db 66 > $O/kernel_stats_cfg2.csv
python tools/latency_summary.py $O/prof2/cfg2_results.db "BASELINE config 2" $O/latency_bound.json
python tools/overlap_stats.py $O/prof2/cfg2_results.db > $O/overlap_cfg2.txt 2>&1
rm -rf $O/prof2
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4j/ab_xp*.json'))+['gpurun_out/r4j/bench_n1.json']:
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f, b['value'], b['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
exit 0
