#!/bin/bash
# round 4, GPU call 7: token-stationary linear kernel at K = 384 (Swin stage 3), A/B at config 2
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g
mkdir -p $O
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed (rc=$rc): stopping"; exit $rc; fi; }
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "test_rowlin" > $O/ktests.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -4 $O/ktests.log; guard $rc
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "swin_t_small or config2_fullsize or boundary or race_free or padded_clip_matches_reference" > $O/e2e.log 2>&1
rc=$?; echo "e2e rc=$rc"; tail -4 $O/e2e.log; guard $rc
B="--no-cpu-baseline --no-roofline --no-variants"
for i in 1 2; do
  TCE_ROWLIN_K384=0 timeout -k 10 200 python bench.py $B > $O/ab_k384_0_$i.json 2> $O/ab.err; guard $?
  TCE_ROWLIN_K384=1 timeout -k 10 200 python bench.py $B > $O/ab_k384_1_$i.json 2> $O/ab.err; guard $?
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4g/ab_k384*.json')):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f, b['value'], b['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 300 python tools/ablate_times.py > $O/ablate.txt 2> $O/ablate.err; head -5 $O/ablate.txt
exit 0
