#!/bin/bash
# round 5, call u: hidden-extent split of the fused FFN at config 3 (40800 rows: the shape with the largest isolated gain), A/B
O=gpurun_out/r6u; mkdir -p $O
for rep in 1 2; do for c in 1 0; do
  TCE_FFN_SPLIT=$c timeout -k 10 300 python bench.py --backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-variants > $O/c3_split${c}_$rep.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/c3_split${c}_$rep.json'));print('cfg3 split=$c', d['value'], d['ms_per_step'])"
done; done
