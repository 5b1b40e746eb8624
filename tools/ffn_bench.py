"""Fused FFN kernel (csrc/chain.hip) vs the two-GEMM + LayerNorm path it replaces: parity vs torch fp64 and time."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="24100x256x2048:relu:out,72000x256x2048:relu:out,18000x256x2048:relu:out,40800x256x2048:relu:out,4600x256x2048:relu:out,"
                                    "72000x96x384:gelu:in,18000x192x768:gelu:in,122880x96x384:gelu:in,256800x128x512:gelu:in")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
torch.manual_seed(0)
dev = "cuda"


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


for spec in a.shapes.split(","):
    shp, actn, ln = spec.split(":")
    M, C, Hd = (int(v) for v in shp.split("x"))
    act = {"relu": 1, "gelu": 2}[actn]
    x = torch.randn(M, C, device=dev)
    w1 = torch.randn(Hd, C, device=dev) / C ** 0.5
    b1 = torch.randn(Hd, device=dev) * 0.1
    w2 = torch.randn(C, Hd, device=dev) / Hd ** 0.5
    b2 = torch.randn(C, device=dev) * 0.1
    g = torch.rand(C, device=dev) + 0.5
    be = torch.randn(C, device=dev) * 0.1
    pk = ops.ffn_pack(w1, b1, w2)
    ln_in = (g, be) if ln == "in" else None
    ln_out = (g, be) if ln == "out" else None
    out = torch.empty_like(x)
    ops.ffn_fused(x, pk, b2, Hd, act, ln_in=ln_in, ln_out=ln_out, out=out)
    torch.cuda.synchronize()
    # fp64 reference on a slice of rows (the kernel treats rows independently)
    idx = torch.cat([torch.arange(0, min(M, 512)), torch.arange(max(0, M - 300), M)]).to(dev)
    xd = x[idx].double()
    y = torch.nn.functional.layer_norm(xd, (C,), g.double(), be.double(), 1e-5) if ln_in else xd
    h = y @ w1.double().t() + b1.double()
    h = torch.relu(h) if act == 1 else torch.nn.functional.gelu(h)
    o = xd + h @ w2.double().t() + b2.double()
    if ln_out:
        o = torch.nn.functional.layer_norm(o, (C,), g.double(), be.double(), 1e-5)
    err = (out[idx].double() - o).abs().max().item()
    scale = o.abs().max().item()

    # the unfused path
    hdn = torch.empty(M, Hd, device=dev)
    xn = torch.empty_like(x)
    x2 = x.clone()

    def unfused():
        src = x2
        if ln_in:
            ops.layernorm(x2, g, be, out=xn)
            src = xn
        ops.gemm_ex(src, w1, hdn, M, Hd, C, C, C, Hd, bias=b1, act=act)
        ops.gemm_ex(hdn, w2, x2, M, C, Hd, Hd, Hd, C, bias=b2, res=x2, ldres=C, res_mode=1)
        if ln_out:
            ops.layernorm(x2, g, be, out=x2)

    t_f = timeit(lambda: ops.ffn_fused(x, pk, b2, Hd, act, ln_in=ln_in, ln_out=ln_out, out=out), a.iters)
    t_u = timeit(unfused, a.iters)
    fl = 4.0 * M * C * Hd
    if C <= 128:  # 256-row workgroups against 128-row ones, two per CU (tce_debug_ffn_set_half)
        from tce_rvos_amd._lib import lib
        res = {}
        for mode in (-1, 1):
            lib().tce_debug_ffn_set_half(mode)
            res[mode] = timeit(lambda: ops.ffn_fused(x, pk, b2, Hd, act, ln_in=ln_in, ln_out=ln_out, out=out), a.iters)
        lib().tce_debug_ffn_set_half(0)
        print(f"{'':32s} 256-row workgroups {res[-1]:8.1f} us   128-row, two per CU {res[1]:8.1f} us")
    nws, ncnt = ops.ffn_split_need(M, C, Hd, act) if (act == 1 and ln_in is None) else (0, 0)
    if nws:  # the hidden-extent split planned for this shape
        ws, cnt = torch.empty(nws, device=dev), torch.zeros(ncnt, dtype=torch.int32, device=dev)
        out_s = torch.empty_like(x)
        t_s = timeit(lambda: ops.ffn_fused(x, pk, b2, Hd, act, ln_out=ln_out, out=out_s, split=(ws, cnt)), a.iters)
        d = (out_s - out).abs().max().item()
        print(f"{'':32s} split: {t_s:8.1f} us = {fl / t_s * 1e-6:6.1f} TFLOP/s alg   x{t_f / t_s:.2f} of the un-split launch   max|split - unsplit| {d:.2e}")
    print(f"{spec:32s} max|err| {err:.2e} (|out| max {scale:.1f})  fused {t_f:8.1f} us = {fl / t_f * 1e-6:6.1f} TFLOP/s alg"
          f"   unfused {t_u:8.1f} us   x{t_u / t_f:.2f}", flush=True)
