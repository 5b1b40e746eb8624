set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u
mkdir -p $O
SH="--shapes 72000x288x96,72000x96x96,72000x256x96"
echo "old build" > $O/rowlin_ab.txt
TCE_LIB=$GRAFT_REPO_ROOT/tce-rvos_amd/lib/ab/libtce_old.so python tools/rowlin_bench.py $SH 2>&1 | grep -v amdgpu >> $O/rowlin_ab.txt
echo "new build (3 workgroups per CU at K = 96)" >> $O/rowlin_ab.txt
python tools/rowlin_bench.py $SH 2>&1 | grep -v amdgpu >> $O/rowlin_ab.txt
cat $O/rowlin_ab.txt
python -m pytest tests/test_kernels_gpu.py -x -q -k "rowlin" > $O/t6.log 2>&1 || { tail -40 $O/t6.log; exit 1; }
tail -2 $O/t6.log
TCE_LIB=$GRAFT_REPO_ROOT/tce-rvos_amd/lib/ab/libtce_old.so python tools/replay_latency.py --reps 100 2>&1 | grep -v amdgpu
python tools/replay_latency.py --reps 100 2>&1 | grep -v amdgpu
TCE_LIB=$GRAFT_REPO_ROOT/tce-rvos_amd/lib/ab/libtce_old.so python tools/replay_latency.py --reps 100 2>&1 | grep -v amdgpu
python tools/replay_latency.py --reps 100 2>&1 | grep -v amdgpu
