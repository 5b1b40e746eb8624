#!/bin/bash
# round 4: HBM-side traffic per kernel on the final build (two counter passes, eager launches)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q
mkdir -p $O
export TCE_GRAPH=0
A="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-variants"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py $A > $O/f.json 2> $O/f.err; rc=$?; echo "fetch rc=$rc"
if [ $rc -ne 0 ]; then tail -5 $O/f.err; exit 1; fi
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py $A > $O/w.json 2> $O/w.err; rc=$?; echo "write rc=$rc"
if [ $rc -ne 0 ]; then tail -5 $O/w.err; exit 1; fi
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
echo $F $W
PMC_OUT=../gpurun_out/r5q/r04_pmc_traffic.json python tools/pmc_traffic.py $F $W 4
ls -la $O/r04_pmc_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5q/r04_pmc_traffic.json"))
print("clip total GB:", d["hbm_bytes_per_clip_all_kernels"]/1e9)
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["hbm_bytes_per_launch"]*kv[1]["launches"])[:12]:
    print(f"{v['hbm_bytes_per_launch']/1e6:9.1f} MB x {v['launches']:4d}  {k[:90]}")
PY
