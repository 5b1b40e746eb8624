set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q
mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -20 $O/bench_n1.err; exit 1; }
python -c "import json;d=json.load(open('$O/bench_n1.json'));print(d['value'],d['ms_per_step'],d['value_c2'],d['value_c4']);print(json.dumps(d['roofline_hbm'],indent=1))"
rocprofv3 --kernel-trace --stats -d $O/prof2 -o cfg2 -- python3 bench.py --steps 40 --no-cpu-baseline --no-roofline --no-variants > $O/bench_under_rocprof.json 2> $O/p2.err
python tools/rocpd_stats.py $O/prof2/cfg2_results.db 46 > $O/kernel_stats_cfg2.csv
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
