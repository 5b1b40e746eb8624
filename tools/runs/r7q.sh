#!/bin/bash
# round 5, call 7q: 3 -> 4 hidden-extent split for fused-FFN launches of more than one round (72000 rows: the stride-4 lateral branch), A/B
# (the TCE_FFN_SPLIT_MULTI switch existed only in the build this script measured: the 3 -> 4 plan for multi-round launches was slower and was removed)
O=gpurun_out/r7q; mkdir -p $O
B="--no-cpu-baseline --no-roofline --no-variants"
for rep in 1 2 3; do for c in 1 0; do
  TCE_FFN_SPLIT_MULTI=$c timeout -k 10 200 python bench.py --steps 200 --warmup 20 $B > $O/c2_multi${c}_$rep.json 2>>$O/err.txt || exit 1
  python -c "import json;d=json.load(open('$O/c2_multi${c}_$rep.json'));print('cfg2 split_multi=$c', d['value'], d['ms_per_step'], d['parity']['max_rel_logit_err'] if d.get('parity') else '')"
done; done
