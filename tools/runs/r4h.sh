#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h
mkdir -p $O
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed (rc=$rc): stopping"; exit $rc; fi; }
timeout -k 10 200 python tools/rowlin384_bench.py > $O/rowlin384.txt 2>&1; rc=$?; cat $O/rowlin384.txt | tail -12; guard $rc
B="--no-cpu-baseline --no-roofline --no-variants"
for i in 1 2; do
  TCE_ROWLIN_K384=0 timeout -k 10 200 python bench.py $B > $O/ab_k384_0_$i.json 2> $O/ab.err; guard $?
  TCE_ROWLIN_K384=1 timeout -k 10 200 python bench.py $B > $O/ab_k384_1_$i.json 2> $O/ab.err; guard $?
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4h/ab_k384*.json')):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f, b['value'], b['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
exit 0
