set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "dynamic_mask_head" > $O/t1.log 2>&1 || { tail -60 $O/t1.log; exit 1; }
tail -2 $O/t1.log
python -m pytest tests/test_e2e_gpu.py -x -q -k "config2 or boundary or aux" > $O/t2.log 2>&1 || { tail -60 $O/t2.log; exit 1; }
tail -2 $O/t2.log
python tools/replay_latency.py --reps 100 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --stats -d $O/prof -o mt4 -- python3 tools/replay_latency.py --reps 30 > $O/prof.log 2>&1
python - <<'PY'
import sqlite3
db=sqlite3.connect('gpurun_out/r3u/prof/mt4_results.db')
for n,c,a in db.execute("select name, count(*), avg(end-start) from kernels group by name"):
    if 'mask_' in n: print(n[:50], c, a)
PY
