#!/bin/bash
# round 5 (final build): kernel traces, counter passes and bench lines on the round's build (profiles/r05_*)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r7u; mkdir -p $O
guard() { rc=$1; if [ $rc -ne 0 ]; then echo "step failed (rc=$rc): stopping"; exit $rc; fi; }
B="--no-cpu-baseline --no-roofline --no-variants"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof2 -o cfg2 -- python3 bench.py --steps 60 $B > $O/bench_cfg2_under_rocprof.json 2> $O/p2.err
rc=$?; echo "prof cfg2 rc=$rc"; guard $rc
python tools/rocpd_stats.py $O/prof2/cfg2_results.db 66 > $O/kernel_stats_cfg2.csv
python tools/latency_summary.py $O/prof2/cfg2_results.db "BASELINE config 2" $O/latency_bound.json
python tools/overlap_stats.py $O/prof2/cfg2_results.db > $O/overlap_cfg2.txt 2>&1
python tools/alone_time.py $O/prof2/cfg2_results.db 40 > $O/alone_cfg2.txt 2>&1
rm -rf $O/prof2
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof2g -o g8 -- python3 bench.py --steps 12 --group 8 $B > $O/bench_cfg2_group8_under_rocprof.json 2> $O/p2g.err
rc=$?; echo "prof cfg2 group8 rc=$rc"; guard $rc
python tools/rocpd_stats.py $O/prof2g/g8_results.db 144 > $O/kernel_stats_cfg2_group8.csv
rm -rf $O/prof2g
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof5 -o cfg5 -- python3 bench.py --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --steps 20 $B > $O/bench_cfg5_under_rocprof.json 2> $O/p5.err
rc=$?; echo "prof cfg5 rc=$rc"; guard $rc
python tools/rocpd_stats.py $O/prof5/cfg5_results.db 26 > $O/kernel_stats_cfg5.csv
rm -rf $O/prof5
A="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-variants"
TCE_GRAPH=0 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py $A > $O/f.json 2> $O/f.err; rc=$?; echo "fetch rc=$rc"; guard $rc
TCE_GRAPH=0 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py $A > $O/w.json 2> $O/w.err; rc=$?; echo "write rc=$rc"; guard $rc
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
PMC_OUT=../gpurun_out/r7u/r05_pmc_traffic.json python tools/pmc_traffic.py $F $W 4 > $O/pmc_traffic.txt; tail -14 $O/pmc_traffic.txt
rm -rf $O/pmc_fetch $O/pmc_write
TCE_GRAPH=0 timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc -o m -- python3 bench.py $A > $O/m.json 2> $O/m.err; rc=$?; echo "mfma pmc rc=$rc"; guard $rc
F=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python tools/pmc_mfma.py $F $O/r05_mfma_util.json > $O/r05_mfma_util.txt; head -16 $O/r05_mfma_util.txt
rm -rf $O/pmc
timeout -k 10 500 python bench.py > $O/bench_n1.json 2> $O/c2.err; rc=$?; echo "cfg2 rc=$rc"; guard $rc
timeout -k 10 400 python bench.py --backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/c3.err; rc=$?; echo "cfg3 rc=$rc"; guard $rc
timeout -k 10 400 python bench.py --backbone resnet50 --frames 1 --no-cpu-baseline > $O/bench_cfg1.json 2> $O/c1.err; rc=$?; echo "cfg1 rc=$rc"; guard $rc
timeout -k 10 400 python bench.py --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --no-cpu-baseline --no-variants > $O/bench_cfg5_f16x3.json 2> $O/c5.err; rc=$?; echo "cfg5 rc=$rc"; guard $rc
timeout -k 10 400 python bench.py --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --arith-policy cfg5_mixed --no-cpu-baseline --no-variants > $O/bench_cfg5_mixed.json 2> $O/c5m.err; rc=$?; echo "cfg5 mixed rc=$rc"; guard $rc
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r7u/bench_*.json')):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); print(f, b['value'], b['ms_per_step'], {k:v for k,v in b.items() if k.startswith('value_')})
    except Exception as e: print(f, 'ERR', e)
print(open('gpurun_out/r7u/latency_bound.json').read())
PY
