"""Swin window attention at the four stages of config 2 (Swin-T, T=5, 360x640) and the last two of config 5 (Swin-B, T=10, 480x854):
split-fp16 kernel (the (1,7,7) form of the 3-D kernel, the default in the split modes) vs the exact-fp32 matrix-core kernel vs the VALU kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops
from tce_rvos_amd._lib import lib
torch.manual_seed(0)
for (T, H, W, nH) in [(5, 90, 160, 3), (5, 45, 80, 6), (5, 23, 40, 12), (5, 12, 20, 24), (10, 30, 54, 16), (10, 15, 27, 32)]:
    C = nH * 32
    qkv = torch.randn(T * H * W, 3 * C, device="cuda"); b = torch.randn(3 * C, device="cuda"); tab = torch.randn(169, nH, device="cuda")
    out = torch.empty(T * H * W, C, device="cuda")
    res = {}
    for mode in (1, 2, 0):
        lib().tce_debug_window_attn_set_mfma(mode)
        for shift in (0, 3):
            for _ in range(3):
                ops.window_attn(qkv, b, tab, T, H, W, C, nH, shift, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.window_attn(qkv, b, tab, T, H, W, C, nH, shift, out=out)
            e1.record(); torch.cuda.synchronize()
            res[(mode, shift)] = e0.elapsed_time(e1) / 20 * 1e3
    lib().tce_debug_window_attn_set_mfma(1)
    mb = T * H * W * 4 * C * 4 / 1e6
    print(f"T={T} {H}x{W} nH={nH}: split-fp16 {res[(1,0)]:6.1f} / {res[(1,3)]:6.1f} us ({mb / res[(1,0)]:.2f} TB/s)   exact-fp32 MFMA {res[(2,0)]:6.1f} / {res[(2,3)]:6.1f} us"
          f"   VALU {res[(0,0)]:6.1f} / {res[(0,3)]:6.1f} us   (plain / shifted)", flush=True)
