set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "fewrow" > $O/t1.log 2>&1 || { tail -40 $O/t1.log; exit 1; }
tail -2 $O/t1.log
python tools/fewrow_bench.py > $O/fewrow_bench.txt 2>&1; cat $O/fewrow_bench.txt
python tools/replay_latency.py > $O/lat_few.txt 2>&1; tail -1 $O/lat_few.txt
TCE_FEWROW_TEXT=0 python tools/replay_latency.py > $O/lat_few_notext.txt 2>&1; tail -1 $O/lat_few_notext.txt
TCE_FEWROW=0 python tools/replay_latency.py > $O/lat_nofew.txt 2>&1; tail -1 $O/lat_nofew.txt
