#!/bin/bash
# round 5, call v: hidden-extent split of the fused FFN with the calibrated plan (2 -> 3 / 1 -> 2 only): tests, e2e, A/B at configs 2 and 3
O=gpurun_out/r6v; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ffn" > $O/k.log 2>&1; rc=$?; echo "kernel rc=$rc"; tail -3 $O/k.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "matches_reference or race_free or taps or replay or video" > $O/e2e.log 2>&1; rc=$?; echo "e2e rc=$rc"; tail -3 $O/e2e.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do for c in 1 0; do
  TCE_FFN_SPLIT=$c timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-variants > $O/b1_split${c}_$rep.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/b1_split${c}_$rep.json'));print('cfg2 B=1 split=$c', d['value'], d['ms_per_step'])"
done; done
for c in 1 0; do
  TCE_FFN_SPLIT=$c timeout -k 10 300 python bench.py --backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-variants > $O/c3_split${c}.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/c3_split${c}.json'));print('cfg3 split=$c', d['value'], d['ms_per_step'])"
done
