#!/bin/bash
# round 5, call f: MFMA utilisation + clock per kernel (one counter pass over an eager clip), and for the exact-fp32 tiled GEMM
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
A="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-variants"
TCE_GRAPH=0 timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc -o m -- python3 bench.py $A > $O/m.json 2> $O/m.err; rc=$?; echo "pmc rc=$rc"
if [ $rc -ne 0 ]; then tail -5 $O/m.err; exit 1; fi
F=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python tools/pmc_mfma.py $F $O/mfma_util_cfg2.json | tee $O/mfma_util_cfg2.txt
rm -rf $O/pmc
cat > /tmp/gemm_f32_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from tce_rvos_amd import ops
ops.set_gemm_mode("f32")
for (M, N, K) in [(24100, 2048, 256), (24100, 256, 2048), (72000, 2048, 256), (4600, 1536, 384)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    out = torch.empty(M, N, device="cuda")
    for _ in range(4):
        ops.gemm(a, w, out=out)
torch.cuda.synchronize()
PY
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc2 -o g -- python3 /tmp/gemm_f32_one.py > $O/g.out 2> $O/g.err; echo "pmc2 rc=$?"
F=$(find $O/pmc2 -name "*counter_collection.csv" | head -1)
python tools/pmc_mfma.py $F $O/mfma_util_gemm_f32.json | tee $O/mfma_util_gemm_f32.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $O/pmc3 -o h -- python3 /tmp/gemm_f32_one.py > $O/h.out 2> $O/h.err; echo "pmc3 rc=$?"
F=$(find $O/pmc3 -name "*counter_collection.csv" | head -1)
python - $F <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_f32" in r["Kernel_Name"]:
        acc[r["Grid_Size"]][r["Counter_Name"]] += float(r["Counter_Value"])
for g, c in acc.items():
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print("grid", g, {k: round(v / wc, 3) for k, v in c.items() if k != "SQ_WAVE_CYCLES"}, "wave_cycles", wc)
PY
rm -rf $O/pmc2 $O/pmc3
