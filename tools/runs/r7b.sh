#!/bin/bash
# round 5, call 7b: counters of the MSDA gather forms (VALU instructions, wait cycles) on tools/msda_bench.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r7b; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o m -- python3 tools/msda_bench.py > $O/m.out 2> $O/m.err; echo "pmc rc=$?"
F=$(find $O/pmc -name "*counter_collection.csv" | head -1); echo $F
python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:40]
    if "msda" not in k: continue
    key = (k, r.get("Grid_Size"))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
for key, v in agg.items():
    n = 1
    print(key, {c: round(x) for c, x in v.items()})
PY
rm -rf $O/pmc
