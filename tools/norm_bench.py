"""Times GroupNorm (stats + apply) and LayerNorm at the sizes of BASELINE config 2."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops
from gemm_bench import bench
for (T, HW, C, G) in [(5, 14400, 256, 8), (5, 3600, 256, 8), (5, 3600, 256, 32)]:
    x = torch.randn(T * HW, C, device="cuda"); g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    out = torch.empty_like(x)
    t = bench(lambda: ops.groupnorm_cl(x, g, b, T, HW, C, G, out=out), 10, graph=True)
    print(f"groupnorm T={T} HW={HW} C={C} G={G}: {t*1e6:7.1f} us  ({3 * x.numel() * 4 / t / 1e12:.2f} TB/s for 2 reads + 1 write)")
for (M, C) in [(24100, 256), (72000, 96), (72000, 256)]:
    x = torch.randn(M, C, device="cuda"); g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    out = torch.empty_like(x)
    t = bench(lambda: ops.layernorm(x, g, b, out=out), 10, graph=True)
    print(f"layernorm {M}x{C}: {t*1e6:7.1f} us  ({2 * x.numel() * 4 / t / 1e12:.2f} TB/s)")
