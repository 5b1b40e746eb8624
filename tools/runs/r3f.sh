set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "fewrow" > $O/t1.log 2>&1 || { tail -40 $O/t1.log; exit 1; }
tail -2 $O/t1.log
python -m pytest tests/test_e2e_gpu.py -x -q > $O/t2.log 2>&1 || { tail -60 $O/t2.log; exit 1; }
tail -2 $O/t2.log
python tools/replay_latency.py > $O/lat_few.txt 2>&1; tail -1 $O/lat_few.txt
TCE_FEWROW=0 python tools/replay_latency.py > $O/lat_nofew.txt 2>&1; tail -1 $O/lat_nofew.txt
TCE_TOKFORK=0 python tools/replay_latency.py > $O/lat_few_notokfork.txt 2>&1; tail -1 $O/lat_few_notokfork.txt
