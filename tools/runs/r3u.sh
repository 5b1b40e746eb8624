set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests_all.log 2>&1 || { tail -40 $O/tests_all.log; exit 1; }
tail -3 $O/tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench_final.json 2> $O/bench_final.err || { tail -20 $O/bench_final.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3u/bench_final.json').read().strip().splitlines()[-1])
print({k:j.get(k) for k in ('value','ms_per_step','value_c2','value_c4','value_text_cached','value_f32_exact')}, j['roofline']['frac'], j['parity']['mask_iou_vs_oracle'])
PY
