#!/bin/bash
# round 5, call 7v: ring depth of the token-stationary linear kernel where one workgroup owns the CU (ROW mode, K > 256): 5 slots against 3
# (two builds: lib/libtce_rvos.so and lib/libtce_rvos_ring3.so = HIPCC_EXTRA=-DROWLIN_RING_DEEP=3, selected with TCE_LIB)
# (the 5-slot build is not in the tree any more: it was slower; rebuild it with -DROWLIN_RING_DEEP=5 on the commit before "deeper rowlin ring measured slower")
O=gpurun_out/r7v; mkdir -p $O
R3=$GRAFT_REPO_ROOT/tce-rvos_amd/lib/libtce_rvos_ring3.so
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rowlin" > $O/k.log 2>&1; rc=$?; echo "kernel rc=$rc"; tail -3 $O/k.log
[ $rc -eq 0 ] || exit 1
for lib in 5 3; do
  if [ $lib = 3 ]; then export TCE_LIB=$R3; else unset TCE_LIB; fi
  timeout -k 10 200 python tools/rowlin384_bench.py 384 4600 > $O/b384_ring$lib.txt 2>>$O/err.txt; echo "ring $lib K=384:"; tail -8 $O/b384_ring$lib.txt
  timeout -k 10 200 python tools/rowlin_bench.py > $O/b256_ring$lib.txt 2>>$O/err.txt; echo "ring $lib K<=256:"; tail -12 $O/b256_ring$lib.txt
done
unset TCE_LIB
B="--no-cpu-baseline --no-roofline --no-variants"
for rep in 1 2; do for lib in 5 3; do
  if [ $lib = 3 ]; then export TCE_LIB=$R3; else unset TCE_LIB; fi
  timeout -k 10 200 python bench.py --steps 200 --warmup 20 $B > $O/c2_ring${lib}_$rep.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/c2_ring${lib}_$rep.json'));print('cfg2 ring=$lib', d['value'], d['ms_per_step'])"
done; done
