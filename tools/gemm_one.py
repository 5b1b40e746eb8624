"""Runs a few launches of chosen GEMM shapes (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops
from tce_rvos_amd._lib import lib
shapes = [(24100, 2048, 256), (72000, 256, 2048), (72000, 384, 96), (24100, 256, 256)]
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib().tce_gemm_force_tile(tile)
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    out = torch.empty(M, N, device="cuda")
    for _ in range(3):
        ops.gemm(a, w, out=out)
torch.cuda.synchronize()
