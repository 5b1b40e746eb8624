set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3v
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python tools/replay_latency.py 2>&1 | tail -1
python tools/msda_bench.py > $O/msda.txt 2>&1 || true; tail -8 $O/msda.txt
