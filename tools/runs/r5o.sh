#!/bin/bash
mkdir -p gpurun_out/r5o
O=gpurun_out/r5o
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "groupnorm" > $O/k.log 2>&1; rc=$?; echo "gn tests rc=$rc"; tail -3 $O/k.log
if [ $rc -ne 0 ]; then exit 1; fi
TCE_GN_SMALL=0 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "groupnorm" > $O/k0.log 2>&1; echo "gn tests (two-launch) rc=$?"; tail -2 $O/k0.log
python - <<'PY' > gpurun_out/r5o/gn_bench.txt 2>&1
import os, sys, torch, subprocess
sys.path.insert(0, os.getcwd())
code = """
import os, sys, torch
sys.path.insert(0, os.getcwd())
from tce_rvos_amd import ops
for (T,HW,G) in [(5,920,32),(5,920,8),(5,240,8),(5,60,32),(5,3600,32)]:
    x=torch.randn(T*HW,256,device='cuda'); ga=torch.ones(256,device='cuda'); be=torch.zeros(256,device='cuda'); out=torch.empty_like(x)
    ws=torch.empty(T*G*400,device='cuda')
    g=torch.cuda.CUDAGraph()
    ops.groupnorm_cl(x,ga,be,T,HW,256,G,relu=True,out=out,ws=ws); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20): ops.groupnorm_cl(x,ga,be,T,HW,256,G,relu=True,out=out,ws=ws)
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"T={T} HW={HW} G={G}: {e0.elapsed_time(e1)/200*1e3:6.2f} us per GroupNorm (chained in a graph)", flush=True)
"""
for env in ("1", "0"):
    print("TCE_GN_SMALL=" + env, flush=True)
    print(subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TCE_GN_SMALL=env), capture_output=True, text=True).stdout, flush=True)
PY
grep -v amdgpu gpurun_out/r5o/gn_bench.txt
B="--no-cpu-baseline --no-roofline --no-variants --steps 80"
for i in 1 2; do
TCE_GN_SMALL=0 timeout -k 10 200 python bench.py $B > $O/cfg2_off_$i.json 2>/dev/null
timeout -k 10 200 python bench.py $B > $O/cfg2_on_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5o/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
PY
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "reference or race_free" > $O/e2e.log 2>&1; echo "e2e rc=$?"; tail -2 $O/e2e.log
