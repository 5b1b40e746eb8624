set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z2
mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -20 $O/bench_n1.err; exit 1; }
rocprofv3 --kernel-trace --stats -d $O/prof2 -o cfg2 -- python3 bench.py --steps 40 --no-cpu-baseline --no-roofline --no-variants > $O/bench_under_rocprof.json 2> $O/p2.err
python tools/rocpd_stats.py $O/prof2/cfg2_results.db 46 > $O/kernel_stats_cfg2.csv
TCE_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-variants > $O/pmc_f.log 2>&1
TCE_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-variants > $O/pmc_w.log 2>&1
PMC_OUT=$O/pmc_traffic.json python tools/pmc_traffic.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv 4 > $O/pmc_sum.txt 2>&1 || { cat $O/pmc_sum.txt; exit 1; }
rm -rf $O/pmc_fetch $O/pmc_write
tail -3 $O/pmc_sum.txt
cat $O/bench_n1.json | cut -c1-600
