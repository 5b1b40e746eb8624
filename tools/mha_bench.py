"""tce_mha_f32 at the pixel decoder's self-attention shapes: split-fp16 kernel (default) against the exact fp32-MFMA kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tce_rvos_amd  # noqa: F401
from tce_rvos_amd import ops
from tce_rvos_amd._lib import lib


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


for (b, nh, Lq, Lk) in ((1, 8, 4600, 4600), (1, 8, 1150, 1150), (5, 8, 3600, 300), (1, 8, 18000, 4600)):
    E = nh * 32
    q, k, v = (torch.randn(b, L, E, device="cuda") for L in (Lq, Lk, Lk))
    out = torch.empty(b, Lq, E, device="cuda")
    f = lambda: ops.mha_core(q, k, v, b, nh, Lq, Lk, E, E, E, Lq * E, Lk * E, Lk * E, out, E, Lq * E)
    A = lambda n: torch.empty(n, dtype=torch.float32, device="cuda")
    ws = A(lib().tce_mha_ws_bytes(b, nh, Lk) // 4)
    t_ws = timeit(lambda: ops.mha_core(q, k, v, b, nh, Lq, Lk, E, E, E, Lq * E, Lk * E, Lk * E, out, E, Lq * E, alloc=lambda n: ws)) \
        if Lk >= ops.MHA_WS_MIN_KEYS else float("nan")
    res = {}
    for on in (1, 0):
        lib().tce_debug_mha_set_split(on)
        res[on] = timeit(f)
    lib().tce_debug_mha_set_split(1)
    fl = 4.0 * b * nh * Lq * Lk * 32
    print(f"batch {b} heads {nh} Lq {Lq} Lk {Lk}: pre-split planes {t_ws:7.1f} us   staged split-fp16 {res[1]:7.1f} us ({fl / res[1] / 1e6:6.1f} TFLOP/s)   fp32 MFMA {res[0]:7.1f} us "
          f"({fl / res[0] / 1e6:6.1f} TFLOP/s)", flush=True)
