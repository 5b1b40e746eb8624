"""Times the Swin PatchEmbed kernel at BASELINE config 2 (5 x 360x640 -> 72000 tokens x 96)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops
from gemm_bench import bench
for (T, H, W, C) in [(5, 360, 640, 96), (10, 480, 854, 128)]:
    x = torch.randn(T, 3, H, W, device="cuda"); w = torch.randn(C, 3, 4, 4, device="cuda") / 7
    b, g, be = (torch.randn(C, device="cuda") for _ in range(3))
    out = torch.empty(T * ((H + 3) // 4) * ((W + 3) // 4), C, device="cuda")
    t = bench(lambda: ops.patch_embed(x, w, b, g, be, out=out), 10, graph=True)
    byts = x.numel() * 4 + out.numel() * 4
    print(f"T={T} {H}x{W} C={C}: {t*1e6:7.1f} us  {byts/t/1e12:5.2f} TB/s algorithmic")
