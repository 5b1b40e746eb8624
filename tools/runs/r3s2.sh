set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s2
mkdir -p $O
timeout -k 10 500 python tools/replay_soak.py --reps 400 > $O/soak_cfg2.txt 2>&1 || { tail -20 $O/soak_cfg2.txt; exit 1; }
tail -2 $O/soak_cfg2.txt
timeout -k 10 500 python tools/replay_soak.py --backbone video_swin_t_p4w7 --frames 8 --height 384 --reps 100 > $O/soak_cfg3.txt 2>&1 || { tail -20 $O/soak_cfg3.txt; exit 1; }
tail -2 $O/soak_cfg3.txt
