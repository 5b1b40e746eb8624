set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u
mkdir -p $O
python -m pytest tests/test_e2e_gpu.py -x -q -s -k "padded_clip" > $O/t2.log 2>&1 || { tail -60 $O/t2.log; exit 1; }
grep -E "padded clip|valid region|passed" $O/t2.log
