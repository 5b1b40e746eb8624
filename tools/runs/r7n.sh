#!/bin/bash
# round 5, call 7n: half-workgroup form of the C <= 128 fused MLP: A/B of the clip at configs 2, 3, 5 (automatic against never)
O=gpurun_out/r7n; mkdir -p $O
B="--no-cpu-baseline --no-roofline --no-variants"
for rep in 1 2; do for c in 0 -1; do
  TCE_FFN_HALF=$c timeout -k 10 200 python bench.py --steps 200 --warmup 20 $B > $O/c2_half${c}_$rep.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/c2_half${c}_$rep.json'));print('cfg2 half=$c', d['value'], d['ms_per_step'])"
done; done
for c in 0 -1; do
  TCE_FFN_HALF=$c timeout -k 10 300 python bench.py --backbone video_swin_t_p4w7 --frames 8 --height 384 --width 640 --steps 100 --warmup 10 $B > $O/c3_half${c}.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/c3_half${c}.json'));print('cfg3 half=$c', d['value'], d['ms_per_step'])"
  TCE_FFN_HALF=$c timeout -k 10 300 python bench.py --backbone swin_b_p4w7 --frames 10 --height 480 --width 854 --steps 40 --warmup 5 $B > $O/c5_half${c}.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/c5_half${c}.json'));print('cfg5 half=$c', d['value'], d['ms_per_step'])"
  TCE_FFN_HALF=$c timeout -k 10 200 python bench.py --steps 40 --warmup 5 --group 8 $B > $O/g8_half${c}.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/g8_half${c}.json'));print('cfg2 G=8 half=$c', d['value'], d['ms_per_step'])"
done
