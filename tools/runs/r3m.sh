set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m
mkdir -p $O
B3="--backbone video_swin_t_p4w7 --frames 8 --height 384"
B5="--backbone swin_b_p4w7 --frames 10 --height 480 --width 854"
python bench.py $B3 --steps 60 --no-variants > $O/bench_cfg3.json 2> $O/cfg3.err; cut -c1-300 $O/bench_cfg3.json
python bench.py $B5 --steps 40 --no-variants > $O/bench_cfg5_f16x3.json 2> $O/cfg5a.err; cut -c1-300 $O/bench_cfg5_f16x3.json
python bench.py $B5 --steps 40 --no-variants --arith-policy cfg5_mixed > $O/bench_cfg5_mixed.json 2> $O/cfg5b.err; cut -c1-300 $O/bench_cfg5_mixed.json
python bench.py $B5 --steps 40 --no-variants --gemm-mode f16 > $O/bench_cfg5_f16.json 2> $O/cfg5c.err; cut -c1-300 $O/bench_cfg5_f16.json
python bench.py --backbone resnet50 --frames 1 --steps 100 --no-variants > $O/bench_cfg1.json 2> $O/cfg1.err; cut -c1-300 $O/bench_cfg1.json
rocprofv3 --kernel-trace --stats -d $O/prof3 -o cfg3 -- python3 bench.py $B3 --steps 20 --no-cpu-baseline --no-roofline --no-variants > $O/p3.json 2> $O/p3.err
python tools/rocpd_stats.py $O/prof3/cfg3_results.db 26 > $O/kernel_stats_cfg3.csv
rocprofv3 --kernel-trace --stats -d $O/prof5 -o cfg5 -- python3 bench.py $B5 --steps 20 --no-cpu-baseline --no-roofline --no-variants --arith-policy cfg5_mixed > $O/p5.json 2> $O/p5.err
python tools/rocpd_stats.py $O/prof5/cfg5_results.db 26 > $O/kernel_stats_cfg5_mixed.csv
head -5 $O/kernel_stats_cfg3.csv | cut -c1-100
