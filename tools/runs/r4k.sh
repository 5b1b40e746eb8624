#!/bin/bash
# round 4, GPU call 11: decoder / resizer launch merges, pack + attention fixes; whole-round A/B; configs 1 / 3 / 5
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k
mkdir -p $O
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed (rc=$rc): stopping"; exit $rc; fi; }
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "xattn or thin or fewrow" > $O/ktests.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -3 $O/ktests.log; guard $rc
[ $rc -ne 0 ] && exit 1
timeout -k 10 500 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "not config5 and not fullsize_configs and not unit_scale" > $O/e2e.log 2>&1
rc=$?; echo "e2e rc=$rc"; tail -3 $O/e2e.log; guard $rc
[ $rc -ne 0 ] && exit 1
B="--no-cpu-baseline --no-roofline --no-variants"
OFF="TCE_SWIN_FUSED=0 TCE_THIN=0 TCE_ROWLIN_K384=0 TCE_XATTN_PACK_FUSED=0"
for i in 1 2; do
  env $OFF timeout -k 10 200 python bench.py $B > $O/ab_round_off_$i.json 2> $O/ab.err; guard $?
  timeout -k 10 200 python bench.py $B > $O/ab_round_on_$i.json 2> $O/ab.err; guard $?
done
timeout -k 10 200 python tools/text_bench.py > $O/text_bench.txt 2>&1; guard $?
timeout -k 10 300 python bench.py --backbone video_swin_t_p4w7 --frames 8 --height 384 --no-cpu-baseline --no-variants > $O/bench_cfg3.json 2> $O/c3.err; guard $?
timeout -k 10 300 python bench.py --backbone resnet50 --frames 1 --no-cpu-baseline --no-variants > $O/bench_cfg1.json 2> $O/c1.err; guard $?
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4k/ab_round*.json'))+['gpurun_out/r4k/bench_cfg3.json','gpurun_out/r4k/bench_cfg1.json']:
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f, b['value'], b['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
tail -4 $O/text_bench.txt
exit 0
