#!/bin/bash
# round 5, call 7z: the schedule's switches re-checked on the final build (each against the default, same call)
O=gpurun_out/r7z; mkdir -p $O
B="--steps 150 --warmup 15 --no-cpu-baseline --no-roofline --no-variants"
run() { tag=$1; shift
  env "$@" timeout -k 10 200 python bench.py $B > $O/$tag.json 2>>$O/err.txt || { echo "$tag failed"; return; }
  python -c "import json;d=json.load(open('$O/$tag.json'));print('$tag', d['value'], d['ms_per_step'])"
}
run default_a X=1
run lat1_enc0 TCE_LAT1_AT=enc0
run lat1_enc1 TCE_LAT1_AT=enc1
run lat1_enc3 TCE_LAT1_AT=enc3
run lat1_backbone TCE_LAT1_AT=backbone
run default_b X=1
run encfork0 TCE_ENCFORK=0
run tokfork0 TCE_TOKFORK=0
run early_proj0 TCE_EARLY_PROJ=0
run text_late0 TCE_TEXT_LATE=0
run ftf_ln_fuse1 TCE_FTF_LN_FUSE=1
run swin3_splitk1 TCE_SWIN3_FC2_SPLITK=1
run default_c X=1
