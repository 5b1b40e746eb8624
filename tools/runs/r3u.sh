set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u
mkdir -p $O
B3="--backbone video_swin_t_p4w7 --frames 8 --height 384 --steps 40 --no-cpu-baseline --no-roofline --no-variants"
TCE_FFN_TAIL_MAX=0 python bench.py $B3 > $O/cfg3_notail.json 2> $O/cfg3.err || { tail -20 $O/cfg3.err; exit 1; }
TCE_FFN_TAIL_FORK=0 python bench.py $B3 > $O/cfg3_tail.json 2> $O/cfg3.err || { tail -20 $O/cfg3.err; exit 1; }
python bench.py $B3 > $O/cfg3_tailfork.json 2> $O/cfg3.err || { tail -20 $O/cfg3.err; exit 1; }
TCE_FFN_TAIL_MAX=0 python bench.py $B3 > $O/cfg3_notail2.json 2> $O/cfg3.err || { tail -20 $O/cfg3.err; exit 1; }
python - <<'PY'
import json
for f in ('cfg3_notail','cfg3_tail','cfg3_tailfork','cfg3_notail2'):
    j=json.loads(open(f'gpurun_out/r3u/{f}.json').read().strip().splitlines()[-1])
    print(f, j['value'], j['ms_per_step'])
PY
