"""Prototype check + timing: pre-split operands + LDS-DMA GEMM (tce_gemm_h2_f32) vs the shipped split-fp16 GEMM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops
from tce_rvos_amd._lib import lib, check
from gemm_bench import bench

def split(x):
    hi = torch.empty(x.shape, dtype=torch.float16, device="cuda"); lo = torch.empty_like(hi)
    check(lib().tce_split_f16_f32(x.data_ptr(), None, hi.data_ptr(), lo.data_ptr(), x.shape[0], x.shape[1], 0, ops._stream()), "split")
    return hi, lo

def h2(ah, al, wh, wl, out, bias=None, act=0, slice_major=False):
    N, K = wh.shape; M = out.shape[0]
    check(lib().tce_gemm_h2_f32(ah.data_ptr(), al.data_ptr(), wh.data_ptr(), wl.data_ptr(), bias.data_ptr() if bias is not None else None,
                                None, out.data_ptr(), M, N, K, 32 if slice_major else K, K, N, 0, act, 0, M * 32 if slice_major else 0,
                                ops._stream()), "h2")


def to_slice_major(p):  # [M, K] -> [K/32, M, 32]
    M, K = p.shape
    return p.view(M, K // 32, 32).permute(1, 0, 2).contiguous()

if __name__ == "__main__":
  for (M, N, K) in [(300, 200, 64), (24100, 2048, 256), (24100, 256, 2048), (72000, 2048, 256), (72000, 256, 2048),
                    (72000, 384, 96), (18000, 768, 192), (4600, 1536, 384), (24100, 256, 256)]:
      g = torch.Generator().manual_seed(M)
      a = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
      ah, al = split(a); wh, wl = split(w)
      out = torch.empty(M, N, device="cuda"); ref = torch.empty(M, N, device="cuda")
      h2(ah, al, wh, wl, out, b, 1)
      ops.gemm(a, w, bias=b, act=ops.ACT_RELU, out=ref)
      torch.cuda.synchronize()
      ref64 = torch.relu(a.double() @ w.double().T + b.double())
      e_h2 = (out.double() - ref64).abs().max().item(); e_ref = (ref.double() - ref64).abs().max().item()
      t_h2 = bench(lambda: h2(ah, al, wh, wl, out, b, 1), 10, graph=True)
      t_ref = bench(lambda: ops.gemm(a, w, bias=b, act=ops.ACT_RELU, out=ref), 10, graph=True)
      ahs, als = to_slice_major(ah), to_slice_major(al)
      out2 = torch.empty(M, N, device="cuda")
      h2(ahs, als, wh, wl, out2, b, 1, slice_major=True)
      torch.cuda.synchronize()
      assert torch.equal(out2, out), "slice-major A must give identical results"
      t_sm = bench(lambda: h2(ahs, als, wh, wl, out2, b, 1, slice_major=True), 10, graph=True)
      t_split = bench(lambda: split(a), 10)
      fl = 2.0 * M * N * K
      print(f"{M:6d}x{N:5d}x{K:5d}  h2 {t_h2*1e6:7.1f}us {fl/t_h2/1e12:6.1f}TF | h2 slice-major A {t_sm*1e6:7.1f}us {fl/t_sm/1e12:6.1f}TF | shipped {t_ref*1e6:7.1f}us {fl/t_ref/1e12:6.1f}TF | "
            f"split(A) {t_split*1e6:6.1f}us | max err h2 {e_h2:.2e} shipped {e_ref:.2e}", flush=True)
