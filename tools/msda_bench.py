"""MSDA fused kernel at the encoder shape of config 2: LDS-staged form vs the L2-gather form."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops
from tce_rvos_amd._lib import lib
torch.manual_seed(0)
for (N, shapes) in [(5, [(45, 80), (23, 40), (12, 20), (6, 10)]), (8, [(48, 80), (24, 40), (12, 20), (6, 10)]),
                    (10, [(60, 107), (30, 54), (15, 27), (8, 14)])]:
    S = sum(h * w for h, w in shapes); Lq = S; M, L, P = 8, 4, 4
    value = torch.randn(N, S, M, 32, device="cuda"); proj = torch.randn(N, Lq, 384, device="cuda")
    ref = torch.rand(Lq, 2, device="cuda"); out = torch.empty(N * Lq, 256, device="cuda")
    res = {}
    for mode in (1, 0):
        lib().tce_debug_msda_set_lds(mode)
        for _ in range(3):
            ops.msda_fused(value, proj, ref, shapes, N, S, M, Lq, L, P, 2, False, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.msda_fused(value, proj, ref, shapes, N, S, M, Lq, L, P, 2, False, out=out)
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 20 * 1e3
    lib().tce_debug_msda_set_lds(0)
    alg = (N * S * 256 * 4 * 2 + N * Lq * 384 * 4) / 1e6
    print(f"N={N} S={S}: LDS-staged {res[1]:7.1f} us ({alg / res[1]:.2f} TB/s algorithmic)   L2 gather {res[0]:7.1f} us   x{res[0] / res[1]:.2f}")
