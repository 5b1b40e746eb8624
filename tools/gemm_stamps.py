"""In-kernel phase stamps of the symmetric split-fp16 GEMM (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops
from tce_rvos_amd._lib import lib
for (M, N, K, tile) in [(24100, 2048, 256, 256128), (72000, 256, 2048, 256128), (72000, 384, 96, 12864), (24100, 256, 256, 12864)]:
    lib().tce_gemm_force_tile(tile)
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    out = torch.empty(M, N, device="cuda")
    for _ in range(3):
        ops.gemm(a, w, out=out)
    buf = torch.zeros(2048 * 8, dtype=torch.int64, device="cuda")
    lib().tce_debug_set_stamp_buffer(buf.data_ptr())
    ops.gemm(a, w, out=out)
    torch.cuda.synchronize()
    lib().tce_debug_set_stamp_buffer(None)
    s = buf.view(2048, 8).cpu()
    s = s[s[:, 0] > 0].double()
    d = lambda i, j: (s[:, j] - s[:, i])
    print(f"{M}x{N}x{K} tile {tile}: blocks sampled {len(s)}")
    for name, i, j in (("prologue", 0, 1), ("k-loop", 1, 2), ("epilogue issue", 2, 3), ("store drain", 3, 4), ("total", 0, 4)):
        x = d(i, j)
        print(f"   {name:15s} median {x.median().item():9.0f} cyc  p10 {x.quantile(0.1).item():9.0f}  p90 {x.quantile(0.9).item():9.0f}")
    span = (s[:, 4].max() - s[:, 0].min()).item()
    print(f"   span of sampled blocks {span:.0f} cyc; k-steps {K // 32}")
lib().tce_gemm_force_tile(0)
