"""Token-stationary linear kernel (csrc/chain.hip) vs the tiled split-fp16 GEMM on the K <= 256 shapes of config 2."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="24100x256x256,24100x384x256,24100x512x256,72000x256x256,72000x288x96,72000x96x96,"
                                    "18000x576x192,18000x192x192,72000x256x96,12000x256x256,4600x256x256")
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
torch.manual_seed(0)


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
    return e0.elapsed_time(e1) / iters * 1e3


for spec in a.shapes.split(","):
    M, N, K = (int(v) for v in spec.split("x"))
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda") * 0.1
    res = torch.randn(M, N, device="cuda")
    g, be = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda") * 0.1
    pk = ops.rowlin_pack(w)
    o1, o2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.rowlin(x, pk, o1, M, N, K, K, N, bias=b, res=res, ldres=N, res_mode=1)
    ops.gemm_ex(x, w, o2, M, N, K, K, K, N, bias=b, res=res, ldres=N, res_mode=1)
    torch.cuda.synchronize()
    ref = (x[:256].double() @ w.double().t() + b.double() + res[:256].double())
    e1_, e2_ = (o1[:256].double() - ref).abs().max().item(), (o2[:256].double() - ref).abs().max().item()
    t_r = timeit(lambda: ops.rowlin(x, pk, o1, M, N, K, K, N, bias=b, res=res, ldres=N, res_mode=1), a.iters)
    t_g = timeit(lambda: ops.gemm_ex(x, w, o2, M, N, K, K, K, N, bias=b, res=res, ldres=N, res_mode=1), a.iters)
    line = f"{spec:18s} rowlin {t_r:7.1f} us ({2.0 * M * N * K / t_r * 1e-6:6.1f} TF/s, err {e1_:.1e})   gemm {t_g:7.1f} us (err {e2_:.1e})   x{t_g / t_r:.2f}"
    if N == 256:
        t_rl = timeit(lambda: ops.rowlin(x, pk, o1, M, N, K, K, N, bias=b, res=res, ldres=N, res_mode=1, ln_out=(g, be)), a.iters)

        def unf():
            ops.gemm_ex(x, w, o2, M, N, K, K, K, N, bias=b, res=res, ldres=N, res_mode=1)
            ops.layernorm(o2, g, be, out=o2)
        t_gl = timeit(unf, a.iters)
        line += f"   | +LN: rowlin {t_rl:7.1f} us  gemm+ln {t_gl:7.1f} us  x{t_gl / t_rl:.2f}"
    print(line, flush=True)
