set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k
mkdir -p $O
for rep in 1 2 3; do python tools/graph_vs_eager.py 2>&1 | grep -v amdgpu | head -1; done
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python tools/replay_latency.py > $O/lat_few.txt 2>&1; tail -1 $O/lat_few.txt
TCE_FEWROW=0 python tools/replay_latency.py > $O/lat_nofew.txt 2>&1; tail -1 $O/lat_nofew.txt
python tools/replay_latency.py > $O/lat_few2.txt 2>&1; tail -1 $O/lat_few2.txt
TCE_FEWROW=0 python tools/replay_latency.py > $O/lat_nofew2.txt 2>&1; tail -1 $O/lat_nofew2.txt
python tools/window_attn3d_bench.py > $O/wa3d.txt 2>&1; cat $O/wa3d.txt
python tools/fewrow_bench.py > $O/fewrow_bench.txt 2>&1; cat $O/fewrow_bench.txt
