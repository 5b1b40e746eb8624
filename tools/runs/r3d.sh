set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d
mkdir -p $O
python tools/replay_latency.py > $O/lat_default.txt 2>&1; tail -1 $O/lat_default.txt
TCE_TEXT_LATE=0 python tools/replay_latency.py > $O/lat_textfirst.txt 2>&1; tail -1 $O/lat_textfirst.txt
TCE_TOKFORK=0 python tools/replay_latency.py > $O/lat_notokfork.txt 2>&1; tail -1 $O/lat_notokfork.txt
TCE_TOKFORK=0 TCE_FORK3=0 python tools/replay_latency.py > $O/lat_nofork3.txt 2>&1; tail -1 $O/lat_nofork3.txt
TCE_EARLY_PROJ=0 python tools/replay_latency.py > $O/lat_noearly.txt 2>&1; tail -1 $O/lat_noearly.txt
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
