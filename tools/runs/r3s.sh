set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python tools/replay_latency.py 2>&1 | tail -1
