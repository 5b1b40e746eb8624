"""Swin-T stage-3 projections (4600 x N x 384, config 2) on the token-stationary kernel vs the tiled GEMM (+ LayerNorm launch).
   python tools/rowlin384_bench.py"""
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
K = 384
for M in (4600, 7680, 18000):
    x = torch.randn(M, K, generator=g).cuda()
    ga, be = torch.ones(K).cuda(), torch.zeros(K).cuda()
    for name, N, ln, act, res in (("norm1->qkv", 1152, True, 0, False), ("proj+res", 384, False, 0, True), ("norm2->fc1+gelu", 1536, True, 2, False)):
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

        print(f"M={M:6d} {name:16s} N={N:5d}: token-stationary {timeit(row):6.1f} us   tiled GEMM{' + LayerNorm' if ln else ''} {timeit(tiled):6.1f} us", flush=True)
