set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests_all.log 2>&1 || { tail -40 $O/tests_all.log; exit 1; }
tail -3 $O/tests_all.log
