#!/bin/bash
# round 4, GPU call 3: the fused Swin attention half-block: kernel tests, micro-benchmark, e2e tests, A/B at config 2
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c
mkdir -p $O
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed (rc=$rc): stopping"; exit $rc; fi; }
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -s -k "swin_attn" > $O/ktests.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -15 $O/ktests.log; guard $rc
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/swin_attn_bench.py > $O/swin_attn_bench.txt 2>&1
rc=$?; cat $O/swin_attn_bench.txt | tail -10; guard $rc
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "swin_t_small or config2_fullsize or boundary or race_free or fullsize_configs_match_reference or padded_clip_matches_reference" > $O/e2e.log 2>&1
rc=$?; echo "e2e rc=$rc"; tail -8 $O/e2e.log; guard $rc
B="--no-cpu-baseline --no-roofline --no-variants"
for i in 1 2; do
  TCE_SWIN_FUSED=0 timeout -k 10 200 python bench.py $B > $O/ab_fused0_$i.json 2> $O/ab.err; guard $?
  TCE_SWIN_FUSED=1 timeout -k 10 200 python bench.py $B > $O/ab_fused1_$i.json 2> $O/ab.err; guard $?
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c/ab_fused*.json')):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f, b['value'], b['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
exit 0
