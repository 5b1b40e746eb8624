"""Swin patch embedding (tce_patch_embed_f32): MFMA kernel (split-fp16 modes) against the fp32 vector kernel (f32 mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tce_rvos_amd  # noqa: F401
from tce_rvos_amd import ops


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (T, H, W, C) in ((5, 360, 640, 96), (5, 360, 640, 128), (32, 480, 854, 96)):
    x = torch.randn(T, 3, H, W, device="cuda")
    w, b = torch.randn(C, 3, 4, 4, device="cuda") / 7, torch.randn(C, device="cuda")
    ga, be = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    ntok = T * ((H + 3) // 4) * ((W + 3) // 4)
    out = torch.empty(ntok, C, device="cuda")
    byt = x.numel() * 4 + ntok * C * 4
    res = {}
    for mode in ("f16x3", "f32"):
        ops.set_gemm_mode(mode)
        res[mode] = timeit(lambda: ops.patch_embed(x, w, b, ga, be, out=out))
    ops.set_gemm_mode("f16x3")
    print(f"T={T} {H}x{W} C={C}: MFMA {res['f16x3']:6.1f} us ({byt / res['f16x3'] / 1e6:5.2f} TB/s)   fp32 vector kernel {res['f32']:6.1f} us "
          f"({byt / res['f32'] / 1e6:5.2f} TB/s)   {byt / 1e6:.1f} MB")
