"""Per-wave phase sums inside the prototype GEMM's K loop (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd._lib import lib
from gemm_h2_bench import split, h2, to_slice_major

for (M, N, K) in [(24100, 2048, 256), (72000, 256, 2048), (24100, 256, 2048), (24100, 256, 256)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    ah, al = split(a); wh, wl = split(w)
    out = torch.empty(M, N, device="cuda")
    for sm in (False, True):
        xa, xl = (to_slice_major(ah), to_slice_major(al)) if sm else (ah, al)
        for _ in range(3):
            h2(xa, xl, wh, wl, out, slice_major=sm)
        buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
        lib().tce_debug_h2_set_stamp_buffer(buf.data_ptr())
        h2(xa, xl, wh, wl, out, slice_major=sm)
        torch.cuda.synchronize()
        lib().tce_debug_h2_set_stamp_buffer(None)
        s = buf.view(-1, 8).cpu().double()
        s = s[s[:, 4] > 0]
        nk = K // 32
        names = ["vmcnt wait", "barrier", "dma issue", "reads+mfma", "k-loop total"]
        print(f"{M}x{N}x{K} {'slice-major' if sm else 'row-major'}: per K slice (ticks of s_memtime), median over {len(s)} waves")
        print("   " + "  ".join(f"{n} {s[:, i].median().item() / nk:7.0f}" for i, n in enumerate(names)))
