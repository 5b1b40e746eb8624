set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python bench.py --steps 150 --no-cpu-baseline --no-roofline --no-variants > $O/c1.json 2> $O/c1.err
python bench.py --steps 100 --clips-in-flight 2 --no-cpu-baseline --no-roofline --no-variants > $O/c2.json 2> $O/c2.err
python bench.py --steps 60 --clips-in-flight 4 --no-cpu-baseline --no-roofline --no-variants > $O/c4.json 2> $O/c4.err
cat $O/c1.json $O/c2.json $O/c4.json
rocprofv3 --kernel-trace --stats -d $O/prof3 -o cfg3 -- python3 bench.py --backbone video_swin_t_p4w7 --frames 8 --height 384 --steps 20 --no-cpu-baseline --no-roofline --no-variants > $O/cfg3.json 2> $O/cfg3.err
rocprofv3 --kernel-trace --stats -d $O/prof5 -o cfg5 -- python3 bench.py --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --steps 20 --no-cpu-baseline --no-roofline --no-variants > $O/cfg5.json 2> $O/cfg5.err
cat $O/cfg3.json $O/cfg5.json
ls $O/prof3 $O/prof5
