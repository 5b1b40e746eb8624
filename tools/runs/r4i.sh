#!/bin/bash
# round 4, GPU call 9: full GPU suite on the round's build, smoke, stride-4 lateral start sweep
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i
mkdir -p $O
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed (rc=$rc): stopping"; exit $rc; fi; }
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log; guard $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?; tail -2 $O/smoke.log; guard $rc
B="--no-cpu-baseline --no-roofline --no-variants"
for at in backbone enc0 enc1 enc2 enc3; do
  TCE_LAT1_AT=$at timeout -k 10 200 python bench.py $B > $O/lat1_$at.json 2> $O/ab.err; guard $?
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4i/lat1_*.json')):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f, b['value'], b['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
exit 0
