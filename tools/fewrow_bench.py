"""Few-row linear kernel vs the tiled matrix-core GEMM on the per-token shapes, inside a replayed graph (50 dependent
launches): us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops

torch.manual_seed(0)


def graph_us(fn, n=50):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 / n * 1e3, g


keep = []
SHAPES = [(40, 256, 256, False), (40, 768, 256, False), (25, 4, 256, False), (40, 386, 256, False),
          (32, 2304, 768, False), (32, 3072, 768, False), (100, 256, 256, False)]
if len(sys.argv) > 1 and sys.argv[1] == "rows":  # clip groups multiply the rows: where does the tiled GEMM take over?
    SHAPES = [(R, N, K, False) for (N, K) in ((256, 256), (768, 256), (2048, 256), (256, 2048)) for R in (25, 40, 100, 160, 200, 256)]
for (R, N, K, ln) in SHAPES:
    x = torch.randn(R, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    res = torch.randn(R, N, device="cuda")
    out = torch.empty(R, N, device="cuda")
    gam, bet = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")

    def few():
        ops.fewrow_linear(x, R, K, [(w, b, out, N, N, False, 0)])

    sk = ops.splitk_for(R, N, K)
    ws = torch.empty(max(1, sk) * R * N, device="cuda")

    def tiled():
        ops.gemm_ex(x, w, out, R, N, K, K, K, N, bias=b, res=res if ln else None, ldres=N, res_mode=ops.RES_ADD if ln else ops.RES_NONE,
                    splitk=sk, ws=ws if sk > 1 else None)
        if ln:
            ops.layernorm(out, gam, bet, 1e-5, out=out)

    a, g1 = graph_us(few)
    c, g2 = graph_us(tiled)
    keep += [g1, g2]
    print(f"R={R:3d} N={N:4d} K={K:4d} {'+res+LN' if ln else '       '}: few-row {a:6.2f} us   tiled GEMM{' (split-K %d)' % sk if sk > 1 else ''}"
          f"{' + LayerNorm' if ln else ''} {c:6.2f} us", flush=True)
