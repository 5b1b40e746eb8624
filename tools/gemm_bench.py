"""GEMM micro-benchmark over the shapes of BASELINE config 2 (tuning aid; run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops
from tce_rvos_amd._lib import lib

SHAPES = [  # (M, N, K, count per clip, tag)
    (24100, 2048, 256, 8, "enc ffn1"), (24100, 256, 2048, 8, "enc ffn2"), (72000, 2048, 256, 1, "vl ffn1 s4"),
    (72000, 256, 2048, 1, "vl ffn2 s4"), (72000, 288, 96, 2, "swin1 qkv"), (72000, 384, 96, 2, "swin1 fc1"),
    (72000, 96, 384, 2, "swin1 fc2"), (18000, 576, 192, 2, "swin2 qkv"), (18000, 768, 192, 2, "swin2 fc1"),
    (18000, 192, 768, 2, "swin2 fc2"), (4600, 1152, 384, 6, "swin3 qkv"), (4600, 1536, 384, 6, "swin3 fc1"),
    (4600, 384, 1536, 6, "swin3 fc2"), (1200, 3072, 768, 2, "swin4 fc1"), (1200, 768, 3072, 2, "swin4 fc2"),
    (24100, 256, 256, 40, "enc 256x256"), (24100, 384, 256, 4, "enc offaw"), (72000, 256, 256, 6, "vl 256 s4"),
    (72000, 160, 256, 1, "mask w0"),
]


def bench(fn, iters=10, graph=False):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if graph:  # GPU-side time of back-to-back launches (host launch cost removed)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f32", "f16x3"]
    tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    if len(sys.argv) > 3:
        lib().tce_debug_set_epilogue(int(sys.argv[3]))
    tot = {}
    for (M, N, K, cnt, tag) in SHAPES:
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        line = f"{tag:12s} {M:6d}x{N:5d}x{K:5d} "
        for mode in modes:
            ops.set_gemm_mode(mode)
            for t in tiles:
                lib().tce_gemm_force_tile(t)
                sec = bench(lambda: ops.gemm(a, w, bias=b, out=out))
                tf = 2.0 * M * N * K / sec / 1e12
                line += f"| {mode}/{t}: {sec*1e6:8.1f}us {tf:7.1f}TF "
                tot[(mode, t)] = tot.get((mode, t), 0.0) + sec * cnt
        print(line, flush=True)
    # 3x3 conv at stride 4
    T, H, W, C = 5, 90, 160, 256
    x = torch.randn(T * H * W, C, device="cuda")
    wc = torch.randn(256, 9 * C, device="cuda") / (9 * C) ** 0.5
    for mode in modes:
        ops.set_gemm_mode(mode)
        for t in tiles:
            lib().tce_gemm_force_tile(t)
            sec = bench(lambda: ops.conv2d_cl(x, wc, T, H, W, C, 3, 3, 1, 1))
            print(f"conv3x3 s4 {mode}/{t}: {sec*1e6:8.1f}us {2.0*T*H*W*256*9*C/sec/1e12:7.1f}TF")
            tot[(mode, t)] += sec * 2
    lib().tce_gemm_force_tile(0)
    print({k: round(v * 1e3, 3) for k, v in tot.items()}, "ms per clip (listed shapes)")


if __name__ == "__main__":
    main()
