set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i
mkdir -p $O
python tools/replay_latency.py > $O/lat_few.txt 2>&1; tail -1 $O/lat_few.txt
TCE_FEWROW=0 python tools/replay_latency.py > $O/lat_nofew.txt 2>&1; tail -1 $O/lat_nofew.txt
python tools/replay_latency.py > $O/lat_few2.txt 2>&1; tail -1 $O/lat_few2.txt
TCE_FEWROW=0 python tools/replay_latency.py > $O/lat_nofew2.txt 2>&1; tail -1 $O/lat_nofew2.txt
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
