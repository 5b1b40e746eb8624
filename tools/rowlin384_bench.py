"""Swin stage-3 projections (Swin-T: 4600 x N x 384 at config 2; Swin-B: 16200 x N x 512 at config 5) on the token-stationary
kernel vs the tiled GEMM (+ LayerNorm launch).
   python tools/rowlin384_bench.py [K [M ...]]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rvos_amd import ops  # noqa: E402


def timeit(fn, n=50):
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


g = torch.Generator().manual_seed(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 384
Ms = [int(a) for a in sys.argv[2:]] or ((4600, 7680, 18000) if K == 384 else (5600, 16200, 24000, 32400))
for M in Ms:
    x = torch.randn(M, K, generator=g).cuda()
    ga, be = torch.ones(K).cuda(), torch.zeros(K).cuda()
    for name, N, ln, act, res in (("norm1->qkv", 3 * K, True, 0, False), ("proj+res", K, False, 0, True), ("norm2->fc1+gelu", 4 * K, True, 2, False)):
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
        b = torch.randn(N, generator=g).cuda()
        pk = ops.rowlin_pack(w)
        out, xn = torch.empty(M, N, device="cuda"), torch.empty_like(x)
        r = torch.randn(M, N, generator=g).cuda() if res else None

        def row():
            ops.rowlin(x, pk, out, M, N, K, K, N, bias=b, act=act, res=r, ldres=N, res_mode=ops.RES_ADD if res else ops.RES_NONE,
                       ln_in=(ga, be) if ln else None)

        def tiled():
            src = x
            if ln:
                ops.layernorm(x, ga, be, out=xn)
                src = xn
            saved, ops._ROUTES.rowlin = ops._ROUTES.rowlin, {}
            ops.gemm_ex(src, w, out, M, N, K, K, K, N, bias=b, act=act, res=r, ldres=N, res_mode=ops.RES_ADD if res else ops.RES_NONE)
            ops._ROUTES.rowlin = saved

        ops.rowlin(x, pk, out, M, N, K, K, N, bias=b, act=act, res=r, ldres=N, res_mode=ops.RES_ADD if res else ops.RES_NONE,
                   ln_in=(ga, be) if ln else None)
        o1 = out.clone()
        tiled()
        err = float((o1 - out).abs().max() / out.abs().max())
        print(f"M={M:6d} {name:16s} N={N:5d}: token-stationary {timeit(row):6.1f} us   tiled GEMM{' + LayerNorm' if ln else ''} {timeit(tiled):6.1f} us   rel diff {err:.1e}", flush=True)
