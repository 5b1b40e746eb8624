set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "window_attention_3d or single_pass or ffn_fused" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python tools/window_attn3d_bench.py > $O/wa3d.txt 2>&1; cat $O/wa3d.txt
python bench.py --backbone video_swin_t_p4w7 --frames 8 --height 384 --steps 40 --no-cpu-baseline --no-roofline --no-variants > $O/cfg3.json 2> $O/cfg3.err; cat $O/cfg3.json
python -m pytest tests/test_e2e_gpu.py -x -q -k "video_swin" > $O/tests2.log 2>&1 || { tail -40 $O/tests2.log; exit 1; }
tail -3 $O/tests2.log
python tools/arith_sensitivity.py --out $O/arith_cfg5.json > $O/arith_cfg5.txt 2>&1 || { tail -30 $O/arith_cfg5.txt; exit 1; }
cat $O/arith_cfg5.txt
python tools/arith_sensitivity.py --unit-scale --out $O/arith_cfg5_unit.json > $O/arith_cfg5_unit.txt 2>&1 || { tail -30 $O/arith_cfg5_unit.txt; exit 1; }
cat $O/arith_cfg5_unit.txt
