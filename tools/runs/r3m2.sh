set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m2
mkdir -p $O
B3="--backbone video_swin_t_p4w7 --frames 8 --height 384"
B5="--backbone swin_b_p4w7 --frames 10 --height 480 --width 854"
python bench.py --backbone resnet50 --frames 1 --steps 100 --no-variants > $O/bench_cfg1.json 2> $O/cfg1.err || { tail -20 $O/cfg1.err; exit 1; }
cut -c1-200 $O/bench_cfg1.json
python bench.py $B3 --steps 60 --no-variants > $O/bench_cfg3.json 2> $O/cfg3.err || { tail -20 $O/cfg3.err; exit 1; }
cut -c1-200 $O/bench_cfg3.json
python bench.py $B5 --steps 40 --no-variants > $O/bench_cfg5_f16x3.json 2> $O/cfg5a.err || { tail -20 $O/cfg5a.err; exit 1; }
cut -c1-200 $O/bench_cfg5_f16x3.json
python bench.py $B5 --steps 40 --no-variants --arith-policy cfg5_mixed > $O/bench_cfg5_mixed.json 2> $O/cfg5b.err || { tail -20 $O/cfg5b.err; exit 1; }
cut -c1-200 $O/bench_cfg5_mixed.json
