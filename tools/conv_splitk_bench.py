"""Level-3 3x3/s2 convolution of C5 (5 x 12x20 x 768 -> 300 rows x 256, K = 6912): plain vs split-K."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops
from gemm_bench import bench
x = torch.randn(5 * 12 * 20, 768, device="cuda"); wc = torch.randn(256, 9 * 768, device="cuda") / 83.0
out = torch.empty(300, 256, device="cuda")
for sk in (1, 2, 4, 8):
    ws = torch.empty(sk * 300 * 256, device="cuda")
    t = bench(lambda: ops.conv2d_cl(x, wc, 5, 12, 20, 768, 3, 3, 2, 1, out=out, splitk=sk, ws=ws if sk > 1 else None), 20, graph=True)
    print(f"splits {sk}: {t*1e6:7.1f} us")
