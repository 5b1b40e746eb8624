"""In-kernel stamps of the rowlin kernel: prologue (x load + fragment build), first-stage wait, tile loop, store drain."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops
from tce_rvos_amd._lib import lib
for (M, N, K) in [(24100, 256, 256), (4600, 256, 256), (72000, 96, 96)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    pk = ops.rowlin_pack(w); out = torch.empty(M, N, device="cuda")
    for _ in range(5):
        ops.rowlin(x, pk, out, M, N, K, K, N, bias=b)
    st = torch.zeros(1024 * 8, dtype=torch.int64, device="cuda")
    lib().tce_debug_ffn_set_stamp_buffer(st.data_ptr())
    ops.rowlin(x, pk, out, M, N, K, K, N, bias=b)
    torch.cuda.synchronize()
    lib().tce_debug_ffn_set_stamp_buffer(None)
    n = min(1024, (M + 127) // 128)
    s = st.view(1024, 8)[:n].double().cpu()
    med = lambda a: a.median().item()
    clk = med((s[:, 3] - s[:, 0]) / ((s[:, 5] - s[:, 4]) * 10e-9)) / 1e9
    print(f"{M}x{N}x{K}: x-frags {med(s[:, 6] - s[:, 0]):7.0f}  first-stage wait {med(s[:, 1] - s[:, 6]):6.0f}  tiles {med(s[:, 2] - s[:, 1]):7.0f} "
          f"({med(s[:, 2] - s[:, 1]) / (N // 32):6.0f}/tile)  drain {med(s[:, 3] - s[:, 2]):6.0f}  total {med(s[:, 3] - s[:, 0]):7.0f} cycles @ {clk:.2f} GHz")
