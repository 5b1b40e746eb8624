set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o
mkdir -p $O
B5="--backbone swin_b_p4w7 --frames 10 --height 480 --width 854"
python bench.py $B5 --steps 30 --no-variants --arith-policy cfg5_mixed > $O/bench_cfg5_mixed.json 2> $O/a.err; python -c "import json;d=json.load(open('$O/bench_cfg5_mixed.json'));print('mixed',d['value'],d['parity'])"
python -m pytest tests/test_e2e_gpu.py -x -q -s -k "config5_mixed or mixed_fp16" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
grep -E "policy|passed" $O/t.log
