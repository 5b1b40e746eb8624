#!/bin/bash
# round 4, GPU call 4: fused Swin attention at two workgroups per CU (A/B of two builds), thin weight-stream layers of RoBERTa
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d
mkdir -p $O
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed (rc=$rc): stopping"; exit $rc; fi; }
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -s -k "swin_attn or thin_linear" > $O/ktests.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -8 $O/ktests.log; guard $rc
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/swin_attn_bench.py > $O/swin_attn_bench.txt 2>&1
rc=$?; tail -8 $O/swin_attn_bench.txt; guard $rc
TCE_LIB=$GRAFT_REPO_ROOT/tools/runs/libtce_alt.so timeout -k 10 200 python tools/swin_attn_bench.py > $O/swin_attn_bench_alt.txt 2>&1
rc=$?; echo "alt build (C=128 at one workgroup per CU):"; tail -8 $O/swin_attn_bench_alt.txt; guard $rc
timeout -k 10 200 python tools/text_bench.py > $O/text_bench.txt 2>&1
rc=$?; tail -4 $O/text_bench.txt; guard $rc
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "text_encoder or swin_t_small or config2_fullsize or boundary or race_free or longer_than_32 or text_cache" > $O/e2e.log 2>&1
rc=$?; echo "e2e rc=$rc"; tail -8 $O/e2e.log; guard $rc
B="--no-cpu-baseline --no-roofline --no-variants"
for i in 1 2; do
  TCE_THIN=0 timeout -k 10 200 python bench.py $B > $O/ab_thin0_$i.json 2> $O/ab.err; guard $?
  TCE_THIN=1 timeout -k 10 200 python bench.py $B > $O/ab_thin1_$i.json 2> $O/ab.err; guard $?
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4d/ab_thin*.json')):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f, b['value'], b['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
exit 0
