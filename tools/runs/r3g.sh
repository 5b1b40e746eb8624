set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -k "fewrow or msda_fused" > $O/t1.log 2>&1 || { tail -40 $O/t1.log; exit 1; }
tail -2 $O/t1.log
python -m pytest tests/test_e2e_gpu.py -x -q -k "text_encoder or swin_t_small or forward_boundary" > $O/t2.log 2>&1 || { tail -60 $O/t2.log; exit 1; }
tail -2 $O/t2.log
python tools/replay_latency.py > $O/lat_few.txt 2>&1; tail -1 $O/lat_few.txt
TCE_FEWROW_TEXT=0 python tools/replay_latency.py > $O/lat_few_notext.txt 2>&1; tail -1 $O/lat_few_notext.txt
rocprofv3 --kernel-trace -d $O/prof -o few -- python3 bench.py --steps 30 --no-cpu-baseline --no-roofline --no-variants > $O/b.json 2> $O/b.err
python tools/rocpd_stats.py $O/prof/few_results.db 36 > $O/stats.csv
grep -E "fewrow|msda_fused|mha_small|mha_mfma|xattn_prepare|ffn_pack" $O/stats.csv | cut -c1-60,150-
