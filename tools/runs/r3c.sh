set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c
mkdir -p $O
python -m pytest tests/test_e2e_gpu.py -x -q -s -k "config5_mixed or mixed_fp16" > $O/tests_e2e.log 2>&1 || { tail -60 $O/tests_e2e.log; exit 1; }
grep -E "policy|passed|failed" $O/tests_e2e.log || true
python -m pytest tests/test_kernels_gpu.py -x -q -s -k "cold" > $O/tests_cold.log 2>&1 || { tail -40 $O/tests_cold.log; exit 1; }
tail -2 $O/tests_cold.log
python tools/graph_node_cost.py > $O/graph_node_cost.txt 2>&1; cat $O/graph_node_cost.txt
rocprofv3 --kernel-trace -d $O/prof_c1 -o c1 -- python3 bench.py --steps 60 --no-cpu-baseline --no-roofline --no-variants > $O/c1.json 2> $O/c1.err
rocprofv3 --kernel-trace -d $O/prof_c2 -o c2 -- python3 bench.py --steps 30 --clips-in-flight 2 --no-cpu-baseline --no-roofline --no-variants > $O/c2.json 2> $O/c2.err
python tools/overlap_stats.py $O/prof_c1/c1_results.db 0.6 > $O/overlap_c1.txt; cat $O/overlap_c1.txt
python tools/overlap_stats.py $O/prof_c2/c2_results.db 0.6 > $O/overlap_c2.txt; cat $O/overlap_c2.txt
GPU_MAX_HW_QUEUES=8 python bench.py --steps 60 --clips-in-flight 2 --no-cpu-baseline --no-roofline --no-variants > $O/c2_q8.json 2> $O/c2_q8.err
GPU_MAX_HW_QUEUES=8 python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-variants > $O/c1_q8.json 2> $O/c1_q8.err
cat $O/c1.json $O/c2.json $O/c2_q8.json $O/c1_q8.json | cut -c1-200
