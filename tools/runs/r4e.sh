#!/bin/bash
# round 4, GPU call 5: where the clip's time goes now (ablation), per-kernel times of the text branch, 3-D window attention
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e
mkdir -p $O
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed (rc=$rc): stopping"; exit $rc; fi; }
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "window_attention_3d" > $O/ktests.log 2>&1
rc=$?; echo "3d kernel tests rc=$rc"; tail -4 $O/ktests.log; guard $rc
timeout -k 10 120 python tools/window_attn3d_bench.py > $O/window_attn3d.txt 2>&1; rc=$?; tail -6 $O/window_attn3d.txt; guard $rc
timeout -k 10 500 python tools/ablate_times.py > $O/ablate.txt 2>&1
rc=$?; cat $O/ablate.txt; guard $rc
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/proft -o text -- python3 tools/text_bench.py > $O/text_prof.log 2>&1
rc=$?; echo "text prof rc=$rc"; guard $rc
python tools/rocpd_stats.py $O/proft/text_results.db > $O/text_kernel_stats.csv; head -12 $O/text_kernel_stats.csv | cut -c1-170
rm -rf $O/proft
exit 0
