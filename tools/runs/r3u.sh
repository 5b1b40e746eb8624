set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u
mkdir -p $O
python -m pytest tests/test_e2e_gpu.py -x -q -k "longer_than_32" > $O/t3.log 2>&1 || { tail -60 $O/t3.log; exit 1; }
tail -3 $O/t3.log
