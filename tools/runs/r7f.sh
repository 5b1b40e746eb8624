#!/bin/bash
# round 5, call 7f: texture-path counters of the MSDA gather (is the vector L1 the bound?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r7f; mkdir -p $O
run() { tag=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/$tag -o m -- python3 tools/msda_bench.py > $O/$tag.out 2> $O/$tag.err; echo "$tag rc=$?"
  F=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    if "msda" not in k: continue
    key = (("q4u" if "q4u" in k else "lds" if "lds" in k else "loop"), r.get("Grid_Size"))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
for key, v in sorted(agg.items()):
    if key[1] == "1544192" or key[0] == "lds": print(key, {c: round(x) for c, x in v.items()})
PY
  rm -rf $O/$tag
}
run a TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
run b TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE
run c SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
