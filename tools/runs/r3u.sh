set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "msda or pos_sine" > $O/t1.log 2>&1 || { tail -40 $O/t1.log; exit 1; }
tail -2 $O/t1.log
python -m pytest tests/test_e2e_gpu.py -x -q -s -k "padded_clip" > $O/t2.log 2>&1 || { tail -60 $O/t2.log; exit 1; }
grep -E "padded clip|passed" $O/t2.log
