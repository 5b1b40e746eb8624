"""Times the one-launch Swin attention half-block (csrc/swinattn.hip) against the three-launch form it replaces, at the
Swin stages of BASELINE configs 2 and 5.   python tools/swin_attn_bench.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rvos_amd import ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    for name, T, H, W, C in (("cfg2 stage0", 5, 90, 160, 96), ("cfg2 stage1", 5, 45, 80, 192), ("cfg5 stage0", 10, 120, 214, 128),
                             ("cfg5 stage1", 10, 60, 107, 256)):
        nH = C // 32
        x = torch.randn(T * H * W, C, generator=g).cuda()
        wqkv, bqkv = (torch.randn(3 * C, C, generator=g) / math.sqrt(C)).cuda(), (torch.randn(3 * C, generator=g) * 0.3).cuda()
        wp, bp = (torch.randn(C, C, generator=g) / math.sqrt(C)).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
        table, g1, b1 = torch.randn(169, nH, generator=g).cuda(), torch.ones(C).cuda(), torch.zeros(C).cuda()
        pk = ops.swin_attn_pack(wqkv, wp)
        ops.rowlin_register(wqkv)
        ops.rowlin_register(wp)
        xn, qkv, att, y = torch.empty_like(x), torch.empty(T * H * W, 3 * C, device="cuda"), torch.empty_like(x), x.clone()

        def three():
            pkq = ops.rowlin_lookup(wqkv, 3 * C, C) if (C <= 128 and T * H * W >= 32768) else None
            if pkq is not None:
                ops.rowlin(x, pkq, qkv, T * H * W, 3 * C, C, C, 3 * C, bias=bqkv, ln_in=(g1, b1))
            else:
                ops.layernorm(x, g1, b1, out=xn)
                ops.gemm_ex(xn, wqkv, qkv, T * H * W, 3 * C, C, C, C, 3 * C, bias=bqkv)
            ops.window_attn(qkv, bqkv, table, T, H, W, C, nH, 3, out=att)
            ops.gemm_ex(att, wp, y, T * H * W, C, C, C, C, C, bias=bp, res=y, ldres=C, res_mode=ops.RES_ADD)

        for shift in (0, 3):
            t_f = timeit(lambda: ops.swin_attn_fused(x, pk, bqkv, bp, table, g1, b1, T, H, W, C, shift, out=y))
            t_3 = timeit(three)
            nwin = T * ((H + 6) // 7) * ((W + 6) // 7)
            print(f"{name}: {T*H*W} tokens C={C} {nwin} windows shift {shift}: fused {t_f:7.1f} us   three launches {t_3:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
