set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u
mkdir -p $O
python -m pytest tests/test_e2e_gpu.py -x -q -k "text_cache or boundary or sharded or rehearsal" > $O/t4.log 2>&1 || { tail -60 $O/t4.log; exit 1; }
tail -3 $O/t4.log
python bench.py --no-cpu-baseline > $O/bench_tc.json 2> $O/bench_tc.err || { tail -20 $O/bench_tc.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3u/bench_tc.json').read().strip().splitlines()[-1])
print({k:j.get(k) for k in ('value','ms_per_step','value_c2','value_c4','value_text_cached','value_f32_exact')})
PY
