#!/bin/bash
# round 4, GPU call 13: how does a clip's time scale with the frame count (a proxy for grouping G clips into one launch program)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m
mkdir -p $O
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed (rc=$rc): stopping"; exit $rc; fi; }
B="--no-cpu-baseline --no-roofline --no-variants --steps 120"
for t in 5 10 15 20; do
  timeout -k 10 200 python bench.py --frames $t $B > $O/frames_$t.json 2> $O/f.err; guard $?
done
python - <<'PY'
import json
base=None
for t in (5,10,15,20):
    b=json.loads(open(f'gpurun_out/r4m/frames_{t}.json').read().strip().splitlines()[-1])
    base = base or b['ms_per_step']
    print(f"T={t:2d}: {b['ms_per_step']:7.3f} ms per clip = {b['ms_per_step']/ (t/5):6.3f} ms per 5 frames ({base*(t/5)/b['ms_per_step']:.3f} x the one-clip-at-a-time rate)")
PY
exit 0
