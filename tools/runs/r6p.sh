#!/bin/bash
# round 5, call p: cross-attention -> FFN chain launch: kernel test, e2e parity with it on, A/B against the two launches
O=gpurun_out/r6p; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "xattn or ffn" > $O/k.log 2>&1; rc=$?; echo "kernel rc=$rc"; tail -5 $O/k.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "matches_reference or race_free or taps or group" > $O/e2e.log 2>&1; rc=$?; echo "e2e rc=$rc"; tail -5 $O/e2e.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do for c in 1 0; do
  TCE_XATTN_FFN_CHAIN=$c timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-variants > $O/b1_chain${c}_$rep.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/b1_chain${c}_$rep.json'));print('B=1 chain=$c', d['value'], d['ms_per_step'], d.get('parity',{}).get('max_rel_logit_err'))"
done; done
for c in 1 0; do
  TCE_XATTN_FFN_CHAIN=$c timeout -k 10 200 python bench.py --steps 40 --warmup 5 --group 8 --no-cpu-baseline --no-roofline --no-variants > $O/g8_chain$c.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/g8_chain$c.json'));print('G=8 chain=$c', d['value'], d['ms_per_step'])"
done
