#!/bin/bash
# round 4, GPU call 6: ablation shares on the current build; text branch with the coalesced thin kernel + row-parallel reduce
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f
mkdir -p $O
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed (rc=$rc): stopping"; exit $rc; fi; }
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "thin_linear or splitk" > $O/ktests.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -4 $O/ktests.log; guard $rc
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "text_encoder or swin_t_small or boundary or longer_than_32" > $O/e2e.log 2>&1
rc=$?; echo "e2e rc=$rc"; tail -4 $O/e2e.log; guard $rc
timeout -k 10 200 python tools/text_bench.py > $O/text_bench.txt 2>&1
rc=$?; tail -4 $O/text_bench.txt; guard $rc
timeout -k 10 600 python tools/ablate_times.py > $O/ablate.txt 2> $O/ablate.err
rc=$?; cat $O/ablate.txt; tail -3 $O/ablate.err; guard $rc
exit 0
