set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "groupnorm or splitk" > $O/t1.log 2>&1 || { tail -40 $O/t1.log; exit 1; }
tail -2 $O/t1.log
B5="--backbone swin_b_p4w7 --frames 10 --height 480 --width 854"
python bench.py $B5 --steps 20 --no-variants --arith-policy cfg5_mixed > $O/bench_cfg5_mixed.json 2> $O/a.err; python -c "import json;d=json.load(open('$O/bench_cfg5_mixed.json'));print('mixed',d['value'],d['parity'])"
python bench.py $B5 --steps 20 --no-variants --arith-policy cfg5_fast > $O/bench_cfg5_fast.json 2> $O/b.err; python -c "import json;d=json.load(open('$O/bench_cfg5_fast.json'));print('fast',d['value'],d['parity'])"
python bench.py $B5 --steps 20 --no-variants --arith-policy cfg5_tight > $O/bench_cfg5_tight.json 2> $O/c.err; python -c "import json;d=json.load(open('$O/bench_cfg5_tight.json'));print('tight',d['value'],d['parity'])"
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python tools/replay_latency.py > $O/lat.txt 2>&1; tail -1 $O/lat.txt
