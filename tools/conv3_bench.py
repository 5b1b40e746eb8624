"""Pixel-stationary 3x3 convolution (tce_conv3x3_f32) against the implicit-GEMM path at the pixel decoder's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tce_rvos_amd  # noqa: F401
from tce_rvos_amd import ops


def timeit(fn, n=20):
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
    g = torch.Generator(device="cpu").manual_seed(0)
    w_cl = (torch.randn(256, 2304, generator=g) / 48.0).cuda()
    pk = ops.conv3x3_pack(w_cl, 256)
    for (T, H, W) in ((5, 90, 160), (5, 45, 80), (8, 96, 160), (10, 120, 214), (40, 90, 160)):
        M = T * H * W
        x = torch.randn(M, 256, generator=g).cuda()
        o1 = torch.empty(M, 256, device="cuda")
        o2 = torch.empty(M, 256, device="cuda")
        t_new = timeit(lambda: ops.conv3x3(x, pk, T, H, W, 256, 256, out=o1))
        from tce_rvos_amd._lib import lib
        forms = {}
        for f in (4, 8):  # one form everywhere (the launcher's own choice above may mix them)
            lib().tce_debug_conv3x3_set_waves(f)
            forms[f] = timeit(lambda: ops.conv3x3(x, pk, T, H, W, 256, 256, out=o1))
        lib().tce_debug_conv3x3_set_waves(0)
        t_new = timeit(lambda: ops.conv3x3(x, pk, T, H, W, 256, 256, out=o1))  # again, warm like the forced forms
        print(f"{'':20s} launcher's choice {t_new:7.1f} us   128-pixel workgroups {forms[4]:7.1f} us   256-pixel workgroups {forms[8]:7.1f} us")
        os.environ["X"] = "1"
        with ops.routes(ops.Routes()):  # nothing registered: the implicit GEMM
            t_old = timeit(lambda: ops.conv2d_cl(x, w_cl, T, H, W, 256, 3, 3, 1, 1, out=o2))
        fl = 2.0 * M * 256 * 2304
        print(f"T={T} {H}x{W} ({M} px): pixel-stationary {t_new:7.1f} us ({fl / t_new / 1e6:6.1f} TFLOP/s alg, "
              f"{3 * fl / t_new / 1e6:6.1f} issued)   implicit GEMM {t_old:7.1f} us   max|d| {(o1 - o2).abs().max().item():.2e}",
              flush=True)


if __name__ == "__main__":
    main()
