"""Video-Swin 3-D window attention: matrix-core kernel (3 x fp16 split) vs the VALU kernel at the four stages of config 3
(Video-Swin-T, T=8, 384x640: windows of 8x7x7 = 392 tokens)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops
from tce_rvos_amd._lib import lib
torch.manual_seed(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for (H, W, nH) in [(96, 160, 3), (48, 80, 6), (24, 40, 12), (12, 20, 24)]:
    C = nH * 32
    qkv = torch.randn(T * H * W, 3 * C, device="cuda"); b = torch.randn(3 * C, device="cuda"); tab = torch.randn(15 * 13 * 13, nH, device="cuda")
    out = torch.empty(T * H * W, C, device="cuda")
    res = {}
    for mode in (1, 0):
        lib().tce_debug_window_attn_set_mfma(mode)
        for shifted in (False, True):
            for _ in range(3):
                ops.window_attn3d(qkv, b, tab, T, H, W, C, nH, shifted, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.window_attn3d(qkv, b, tab, T, H, W, C, nH, shifted, out=out)
            e1.record(); torch.cuda.synchronize()
            res[(mode, shifted)] = e0.elapsed_time(e1) / 10 * 1e3
    lib().tce_debug_window_attn_set_mfma(1)
    nwin = ((H + 6) // 7) * ((W + 6) // 7) * ((T + 7) // 8)
    fl = 4.0 * nwin * nH * min(T, 8) * 49 * min(T, 8) * 49 * 32
    print(f"T={T} {H}x{W} nH={nH} ({nwin * nH} workgroups): MFMA {res[(1,False)]:7.1f} / {res[(1,True)]:7.1f} us "
          f"({fl / res[(1,False)] / 1e6:6.1f} TFLOP/s alg)   VALU {res[(0,False)]:7.1f} / {res[(0,True)]:7.1f} us   (plain / shifted)",
          flush=True)
