"""Small / latency-bound GEMM shapes of the path (decoder, frame tokens, text K/V, level-3 conv)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops
from tce_rvos_amd._lib import lib
from gemm_bench import bench

SHAPES = [(25, 256, 256), (40, 512, 256), (32, 256, 256), (25, 2048, 256), (25, 256, 2048), (100, 2153, 256),
          (300, 256, 256), (1500, 256, 256), (1500, 2048, 256), (1500, 256, 2048), (1100, 512, 256), (4600, 256, 384)]
for tile in (6464, 6465):
    lib().tce_gemm_force_tile(tile)
    tot = 0
    for (M, N, K) in SHAPES:
        a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
        sec = bench(lambda: ops.gemm(a, w, out=out), 20, graph=True)
        tot += sec
        print(f"tile {tile} {M:5d}x{N:5d}x{K:5d}: {sec*1e6:7.1f} us")
    x = torch.randn(5 * 12 * 20, 768, device="cuda"); wc = torch.randn(256, 9 * 768, device="cuda")
    co = torch.empty(5 * 6 * 10, 256, device='cuda')
    sec = bench(lambda: ops.conv2d_cl(x, wc, 5, 12, 20, 768, 3, 3, 2, 1, out=co), 20, graph=True)
    print(f"tile {tile} conv C5 3x3s2: {sec*1e6:7.1f} us; sum plain {tot*1e6:.1f} us")
lib().tce_gemm_force_tile(0)
