"""Probe: the frame-token chain (few-row projections, attention core, LayerNorm) replayed in a graph NEXT TO big launches
on a second branch -- are its results the serial ones?  (bisecting a replay-vs-eager mismatch, round 3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops

torch.manual_seed(0)
R, D = 40, 256
dev = "cuda"
tok0 = torch.randn(R, D, device=dev)
tpos = torch.randn(8, D, device=dev)
Wqkv = torch.randn(768, D, device=dev) / 16
bqkv = torch.randn(768, device=dev) * 0.1
Wo = torch.randn(D, D, device=dev) / 16
bo = torch.randn(D, device=dev) * 0.1
g1, b1 = torch.ones(D, device=dev), torch.zeros(D, device=dev)
token = tok0.clone()
qk, v, att, k2, v2 = (torch.empty(R, n, device=dev) for n in (512, 256, 256, 256, 256))
# big work for the other branch
M = 72000
xb = torch.randn(M, 256, device=dev)
w1 = torch.randn(2048, 256, device=dev) / 16
w2 = torch.randn(256, 2048, device=dev) / 45
pkb = ops.ffn_pack(w1, torch.zeros(2048, device=dev), w2)
b2 = torch.zeros(256, device=dev)
outb = torch.empty_like(xb)
outb2 = torch.empty_like(xb)
wsq = torch.randn(256, 256, device=dev) / 16
gn_g, gn_b = torch.ones(256, device=dev), torch.zeros(256, device=dev)
USE_FEW = os.environ.get("PROBE_FEW", "1") == "1"


def chain(n):
    for _ in range(n):
        if USE_FEW:
            ops.fewrow_linear(token, R, D, [(Wqkv[:512], bqkv[:512], qk, 512, 512, True, 0), (Wqkv[512:], bqkv[512:], v, 256, 256, False, 0)],
                              a2=tpos, lda2=D, a2_rows=8)
        else:
            ops.gemm_ex(token, Wqkv[:512], qk, 8, 512, D, D, D, 512, bias=bqkv[:512], a2=tpos, lda2=D, batch=5, sA=8 * D, sA2=0, sC=8 * 512)
            ops.gemm_ex(token, Wqkv[512:], v, R, 256, D, D, D, 256, bias=bqkv[512:])
        ops.mha_core(qk, qk[:, D:], v, 1, 8, R, R, 512, 512, 256, 0, 0, 0, att, 256, 0)
        if USE_FEW:
            ops.fewrow_linear(att, R, D, [(Wo, bo, token, D, D, False, 0)], res=token, ldres=D)
        else:
            ops.gemm_ex(att, Wo, token, R, D, D, D, D, D, bias=bo, res=token, ldres=D, res_mode=ops.RES_ADD)
        ops.layernorm(token, g1, b1, 1e-5, out=token)
        if USE_FEW:
            ops.fewrow_linear(token, R, D, [(Wqkv[256:512], bqkv[256:512], k2, 256, 256, True, 0), (Wqkv[512:], bqkv[512:], v2, 256, 256, False, 0)],
                              a2=tpos, lda2=D, a2_rows=8)
        else:
            ops.gemm_ex(token, Wqkv[256:512], k2, R, 256, D, D, D, 256, bias=bqkv[256:512])
            ops.gemm_ex(token, Wqkv[512:], v2, R, 256, D, D, D, 256, bias=bqkv[512:])


def run(with_big, n=24):
    token.copy_(tok0)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    chain(1)
    token.copy_(tok0)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        if with_big:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(6):
                    if os.environ.get("PROBE_BIG", "mix") == "ffn":
                        ops.ffn_fused(xb, pkb, b2, 2048, ops.ACT_RELU, out=outb)
                    else:  # the kinds of launches a lateral branch of the pixel decoder makes
                        ops.gemm_ex(xb, wsq, outb, M, 256, 256, 256, 256, 256, bias=b2)
                        ops.groupnorm_cl(outb, gn_g, gn_b, 5, M // 5, 256, 8, out=outb2)
                        ops.layernorm(outb2, gn_g, gn_b, 1e-5, out=outb)
                        ops.gemm_ex(outb, wsq[:, :], outb2, M, 256, 256, 256, 256, 256, bias=b2, res=outb, ldres=256, res_mode=ops.RES_ADD)
        chain(n)
        if with_big:
            cur.wait_stream(side)
    res = []
    for _ in range(4):
        token.copy_(tok0)
        g.replay()
        torch.cuda.synchronize()
        res.append((token.clone(), k2.clone()))
    return res, g


serial, g0 = run(False)
conc, g1_ = run(True)
print("serial replays identical:", all(torch.equal(serial[0][0], s[0]) for s in serial))
for i, (t, k) in enumerate(conc):
    print(f"concurrent replay {i}: max|token - serial| {(t - serial[0][0]).abs().max().item():.3e}  max|k - serial| {(k - serial[0][1]).abs().max().item():.3e}")
