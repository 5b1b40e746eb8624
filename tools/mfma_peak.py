"""Chip-wide sustained fp16 MFMA rate under the split GEMM's instruction mix (no memory system): the practical ceiling
below the nominal 2.5 PFLOP/s once the whole chip issues MFMAs (clock / power limited)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd._lib import lib
from tce_rvos_amd import ops
L = lib()
out = torch.empty(256 * 8 * 512, device="cuda")
for blocks, threads, lds in [(256, 256, 0), (256, 512, 0), (256, 256, 1), (256, 512, 1), (256, 512, 2), (256, 512, 3), (190, 512, 2), (190, 512, 3), (64, 512, 2)]:
    iters = 4000
    for _ in range(2):
        L.tce_debug_mfma_peak(out.data_ptr(), blocks, threads, iters, lds, ops._stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.tce_debug_mfma_peak(out.data_ptr(), blocks, threads, iters, lds, ops._stream())
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 5
    nm = blocks * (threads // 64) * iters * 12
    fl = nm * 32 * 32 * 16 * 2
    per_simd = blocks * (threads // 64) / 4.0 / min(blocks, 256)   # waves per SIMD
    cyc = t / (iters * 12 * max(1, (threads // 64) / 4))            # seconds per MFMA per SIMD
    print(f"blocks {blocks} threads {threads} lds {lds}: {t*1e3:7.3f} ms  {fl/t/1e12:8.1f} TFLOP/s  -> {1/cyc/1e9*32:6.2f} GHz-equivalent (32 cyc/MFMA)", flush=True)
