"""Times tce_gemm_f32 at given shapes (MxNxK ...), per launch inside a replayed graph of 20 chained launches;
ALL_TILES=1 also forces each tile (tce_gemm_force_tile) -- used for tile-selection decisions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tce_rvos_amd  # noqa: F401
from tce_rvos_amd import ops
from tce_rvos_amd._lib import lib

shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4600, 1152, 384), (4600, 384, 384), (4600, 1536, 384),
                                                                         (4600, 384, 1536), (1200, 2304, 768), (1200, 768, 768),
                                                                         (1200, 3072, 768), (1200, 768, 3072), (18000, 576, 192)]
TILES = [0, 256128, 12864, 6464] if os.environ.get("ALL_TILES") else [0]


def time_shape(M, N, K, tile):
    lib().tce_gemm_force_tile(tile)
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5, torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    g = torch.cuda.CUDAGraph()
    ops.gemm_ex(x, w, out, M, N, K, K, K, N, bias=b)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20):
            ops.gemm_ex(x, w, out, M, N, K, K, K, N, bias=b)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 100 * 1e3


if os.environ.get("BENCH_GEMM_MODE"):  # f16: single-pass arithmetic (how much of a shape's time is MFMA / split work?)
    ops.set_gemm_mode(os.environ["BENCH_GEMM_MODE"])
_r = ops.routes(ops.Routes())  # nothing registered: the tiled GEMM
_r.__enter__()
for M, N, K in shapes:
    lib().tce_gemm_force_tile(0)  # (a forced tile of the previous shape would be echoed back)
    sel = lib().tce_gemm_select_tile_ex(M, N, K, 1, 0)
    res = {t: time_shape(M, N, K, t) for t in TILES}
    print(f"{M}x{N}x{K}: selected {sel}: " + "  ".join(f"{'auto' if t == 0 else t}: {us:6.1f} us" for t, us in res.items()) +
          f"   ({2.0 * M * N * K / res[0] / 1e6:6.1f} TFLOP/s auto)", flush=True)
lib().tce_gemm_force_tile(0)
_r.__exit__()
