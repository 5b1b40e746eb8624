"""MSDA fused kernel at the encoder shapes of configs 2 / 3 / 5: the forms of the 16-byte gather (tce_debug_msda_set_lds: 2 = one
point at a time, 3 / 4 = two / four points in flight, 1 = LDS-staged coarse levels)."""
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
    res, outs = {}, {}
    for mode in (2, 3, 4, 1):
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
        outs[mode] = out.clone()
    lib().tce_debug_msda_set_lds(0)
    alg = (N * S * 256 * 4 * 2 + N * Lq * 384 * 4) / 1e6
    same = all(torch.equal(outs[2], outs[k]) for k in (3, 4, 1))
    print(f"N={N} S={S}: one at a time {res[2]:7.1f} us   two in flight {res[3]:7.1f} us   four in flight {res[4]:7.1f} us "
          f"({alg / res[4]:.2f} TB/s algorithmic)   LDS-staged {res[1]:7.1f} us   equal results: {same}", flush=True)
