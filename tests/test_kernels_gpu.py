"""GPU: every HIP kernel (through the C ABI) against the CPU oracle / plain torch fp32 on seeded inputs."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import tce_oracle as O  # noqa: E402
from _util import load_npz  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from tce_rvos_amd import ops as _ops
    return _ops


def dev(t):
    return t.cuda().contiguous()


def close(a, b, rtol=1e-4, atol=1e-4):
    a = a.detach().cpu()
    b = b.detach().cpu()
    d = (a - b).abs().max().item()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"max abs diff {d}"


@pytest.mark.parametrize("M,N,K", [(1, 1, 16), (25, 256, 256), (130, 70, 48 * 2), (300, 384, 96), (1200, 2048, 256),
                                   (4097, 96, 384), (513, 2153, 256), (3333, 1000, 160), (24100, 256, 64),
                                   (2049, 1500, 32)])
def test_gemm_plain(ops, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    out = ops.gemm(dev(a), dev(w), bias=dev(b))
    close(out, F.linear(a, w, b), 1e-4, 1e-4)


@pytest.mark.parametrize("M,N,K", [(1024, 512, 2048), (300, 96, 384)])
def test_gemm_split_fp16_is_fp32_accurate(ops, M, N, K):
    """Both GEMM arithmetic modes against an fp64 reference: the 3 x fp16 split must stay within a small factor of
    the exact-fp32 MFMA kernel's error (and far inside what separates fp16/bf16 GEMMs from fp32)."""
    g = torch.Generator().manual_seed(11)
    a = torch.randn(M, K, generator=g) * torch.logspace(-2, 2, K)[None, :]   # 4 decades of dynamic range along K
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    ref = (a.double() @ w.double().T)
    scale = (a.double().abs() @ w.double().abs().T)   # sum |a||b|: the natural error scale
    errs = {}
    try:
        for mode in ("f32", "f16x3"):
            ops.set_gemm_mode(mode)
            out = ops.gemm(dev(a), dev(w)).cpu().double()
            errs[mode] = ((out - ref).abs() / scale).max().item()
    finally:
        ops.set_gemm_mode("f16x3")
    print("relative-to-sum|a||b| errors:", errs)
    assert errs["f32"] < 2e-6
    assert errs["f16x3"] < 2e-6 and errs["f16x3"] < 8 * max(errs["f32"], 1e-7)


def test_gemm_epilogues_and_prologue(ops):
    g = torch.Generator().manual_seed(3)
    M, N, K = 777, 200, 64
    a, a2 = torch.randn(M, K, generator=g), torch.randn(M, K, generator=g)
    w, b, r = torch.randn(N, K, generator=g) / 8, torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    close(ops.gemm(dev(a), dev(w), bias=dev(b), a2=dev(a2)), F.linear(a + a2, w, b))
    close(ops.gemm(dev(a), dev(w), bias=dev(b), act=ops.ACT_RELU), F.relu(F.linear(a, w, b)))
    close(ops.gemm(dev(a), dev(w), bias=dev(b), act=ops.ACT_GELU), F.gelu(F.linear(a, w, b)))
    close(ops.gemm(dev(a), dev(w), bias=dev(b), res=dev(r), res_mode=ops.RES_ADD), F.linear(a, w, b) + r)
    close(ops.gemm(dev(a), dev(w), bias=dev(b), res=dev(r), res_mode=ops.RES_MUL), F.linear(a, w, b) * r)
    close(ops.gemm(dev(a), dev(w)), F.linear(a, w))
    # row-strided views: A is a column slice of a wider buffer, out likewise
    wide = torch.randn(M, 3 * K, generator=g)
    dw = dev(wide)
    outw = torch.zeros(M, 2 * N, device="cuda")
    ops.gemm(dw[:, K:2 * K], dev(w), out=outw[:, N:])
    close(outw[:, N:], F.linear(wide[:, K:2 * K], w))
    assert outw[:, :N].abs().sum().item() == 0


def test_gemm_large_epilogues_shared_addend(ops):
    """Large problem (persistent producer/consumer kernel): frame-batched launch with a stride-0 shared addend,
    bias + ReLU + residual, output written into a level slice of a wider [T, S, C] buffer."""
    g = torch.Generator().manual_seed(8)
    T, hw, S, K, N = 5, 3600, 4820, 256, 256
    a, pos = torch.randn(T, hw, K, generator=g), torch.randn(hw, K, generator=g)
    w, b = torch.randn(N, K, generator=g) / 16, torch.randn(N, generator=g)
    r = torch.randn(T, hw, N, generator=g)
    out = torch.zeros(T, S, N, device="cuda")
    ops.gemm_ex(dev(a), dev(w), out[:, 100:], hw, N, K, K, K, N, bias=dev(b), a2=dev(pos), lda2=K, act=ops.ACT_RELU,
                res=dev(r), ldres=N, res_mode=ops.RES_MUL, batch=T, sA=hw * K, sA2=0, sC=S * N, sRes=hw * N)
    ref = torch.relu(F.linear(a + pos[None], w, b)) * r
    close(out[:, 100:100 + hw], ref, 1e-4, 1e-4)
    assert out[:, :100].abs().sum().item() == 0 and out[:, 100 + hw:].abs().sum().item() == 0


@pytest.mark.parametrize("B,M,N,K", [(3, 333, 160, 256), (5, 4820, 384, 256)])
def test_gemm_batched(ops, B, M, N, K):
    g = torch.Generator().manual_seed(4)
    a, w = torch.randn(B, M, K, generator=g), torch.randn(B, N, K, generator=g) / 16
    out = torch.empty(B, M, N, device="cuda")
    ops.gemm_batched(dev(a), dev(w), out)
    close(out, torch.bmm(a, w.transpose(1, 2)))


@pytest.mark.parametrize("T,H,W,Cin,N,k,s,p", [(2, 9, 13, 32, 48, 3, 1, 1), (3, 12, 20, 64, 256, 3, 2, 1),
                                                (1, 23, 40, 256, 256, 3, 1, 1), (2, 7, 5, 16, 33, 1, 1, 0),
                                                (4, 45, 81, 64, 250, 3, 1, 1), (3, 90, 61, 32, 256, 3, 2, 1)])
def test_conv_implicit_gemm(ops, T, H, W, Cin, N, k, s, p):
    g = torch.Generator().manual_seed(T + H + N)
    x = torch.randn(T, Cin, H, W, generator=g)
    w = torch.randn(N, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(N, generator=g)
    ref = F.conv2d(x, w, b, stride=s, padding=p)
    x_cl = x.permute(0, 2, 3, 1).reshape(T * H * W, Cin)
    w_p = w.permute(0, 2, 3, 1).reshape(N, k * k * Cin)
    out, Ho, Wo = ops.conv2d_cl(dev(x_cl), dev(w_p), T, H, W, Cin, k, k, s, p, bias=dev(b))
    assert (Ho, Wo) == tuple(ref.shape[-2:])
    close(out.view(T, Ho, Wo, N).permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("M,C", [(5, 96), (1000, 256), (77, 768), (33, 3072), (9, 4), (72001, 96), (1003, 128), (17, 132), (24101, 256), (50, 192), (7, 260)])
def test_layernorm(ops, M, C):
    g = torch.Generator().manual_seed(M + C)
    x, r = torch.randn(M, C, generator=g) * 3 + 1, torch.randn(M, C, generator=g)
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    close(ops.layernorm(dev(x), dev(ga), dev(be)), F.layer_norm(x, (C,), ga, be), 1e-4, 1e-5)
    close(ops.layernorm(dev(x), dev(ga), dev(be), r=dev(r), eps=1e-12), F.layer_norm(x + r, (C,), ga, be, 1e-12), 1e-4, 1e-5)


@pytest.mark.parametrize("T,HW,C,G,relu", [(2, 60, 256, 32, False), (3, 1300, 256, 8, True), (1, 14400, 64, 8, True),
                                           (5, 14400, 256, 8, True), (2, 4097, 256, 32, False), (1, 513, 256, 64, False),
                                           (1, 7, 256, 16, True),
                                           # small maps, and data with a large common offset (below)
                                           (5, 920, 256, 32, False), (5, 920, 256, 8, True), (5, 240, 256, 8, True),
                                           (2, 1024, 256, 32, True), (1, 1025, 256, 32, True), (3, 63, 256, 16, False)])
def test_groupnorm(ops, T, HW, C, G, relu):
    g = torch.Generator().manual_seed(HW)
    x = torch.randn(T, HW, C, generator=g) * 2 + 0.5
    if HW in (240, 1024):  # a large common offset: the statistics must not be a sum / sum-of-squares difference
        x = x * 0.05 + 40.0
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    # fp64 reference (torch's fp32 GroupNorm on the CPU is a sum / sum-of-squares: off by 0.8 on the offset data)
    ref = F.group_norm(x.double().permute(0, 2, 1).reshape(T, C, HW, 1), G, ga.double(), be.double(), 1e-5).float()
    if relu:
        ref = F.relu(ref)
    out = ops.groupnorm_cl(dev(x.reshape(T * HW, C)), dev(ga), dev(be), T, HW, C, G, relu=relu)
    tol = 2e-3 if HW in (240, 1024) else 1e-4   # offset data: |x| / std = 400, so fp32 input rounding alone is 400 * 6e-8 * 3 sigma
    close(out.view(T, HW, C).permute(0, 2, 1).reshape(T, C, HW, 1), ref, tol, tol * 0.1)


@pytest.mark.parametrize("T,h,w,ho,wo,G", [(5, 45, 80, 90, 160, 8), (5, 23, 40, 45, 80, 8), (5, 12, 20, 23, 40, 8), (2, 6, 7, 13, 15, 32),
                                           (1, 3, 3, 3, 3, 8)])
def test_groupnorm_up_add_is_the_two_launches(ops, T, h, w, ho, wo, G):
    """tce_groupnorm_up_add_f32 (round 5): the GroupNorm + ReLU of a coarse map applied WHILE it is up-sampled (nearest, to an
    exact finer size: odd sizes, non-integer ratios) and added to the finer map -- the pixel decoder's top-down merge -- against
    tce_groupnorm_f32 followed by tce_resize_nearest_f32: same arithmetic, bit-identical; and against torch; in place on add."""
    C = 256
    g = torch.Generator().manual_seed(h * w + ho)
    x = torch.randn(T * h * w, C, generator=g) * 2 + 0.5
    fine = torch.randn(T * ho * wo, C, generator=g)
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    y = ops.groupnorm_cl(dev(x), dev(ga), dev(be), T, h * w, C, G, relu=True)
    two = ops.resize_nearest(y, T, h, w, ho, wo, C, add=dev(fine))
    one = dev(fine)
    ops.groupnorm_up_add(dev(x), dev(ga), dev(be), T, h, w, ho, wo, C, G, add=one, out=one, relu=True)
    assert torch.equal(one, two)
    ref = F.relu(F.group_norm(x.double().view(T, h * w, C).permute(0, 2, 1).reshape(T, C, h, w), G, ga.double(), be.double(), 1e-5))
    ref = fine.double().view(T, ho, wo, C) + F.interpolate(ref, size=(ho, wo), mode="nearest").permute(0, 2, 3, 1)
    close(one.view(T, ho, wo, C), ref.float(), 1e-4, 1e-4)


@pytest.mark.parametrize("T,h,w,ho,wo", [(5, 12, 20, 90, 160), (5, 12, 20, 45, 80), (2, 12, 20, 23, 40), (3, 5, 7, 11, 13), (1, 1, 1, 3, 2)])
def test_resize_bilinear_ln_is_the_two_launches(ops, T, h, w, ho, wo):
    """tce_resize_bilinear_ln_f32 (round 5): LayerNorm(add + bilinear up-sampling) as one pass against tce_resize_bilinear_f32 then
    tce_layernorm_f32 -- the same operation sequence up to the compiler's choice of fused multiply-adds in the two kernels'
    LayerNorm arithmetic: equal to a few ulp -- and against torch (align_corners=False); in place on add."""
    C = 256
    g = torch.Generator().manual_seed(h * w + ho)
    low = torch.randn(T * h * w, C, generator=g)
    fine = torch.randn(T * ho * wo, C, generator=g) * 2 + 0.3
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    two = ops.resize_bilinear(dev(low), T, h, w, ho, wo, C, add=dev(fine))
    two = ops.layernorm(two, dev(ga), dev(be), out=two)
    one = dev(fine)
    ops.resize_bilinear(dev(low), T, h, w, ho, wo, C, add=one, out=one, ln=(dev(ga), dev(be)))
    assert float((one - two).abs().max()) <= 4e-6 * float(two.abs().max())
    up = F.interpolate(low.double().view(T, h, w, C).permute(0, 3, 1, 2), size=(ho, wo), mode="bilinear", align_corners=False)
    ref = F.layer_norm(fine.double().view(T, ho, wo, C) + up.permute(0, 2, 3, 1), (C,), ga.double(), be.double(), 1e-5)
    close(one.view(T, ho, wo, C), ref.float(), 1e-4, 1e-4)


@pytest.mark.parametrize("T,H,W,C", [(2, 72, 100, 96), (1, 30, 41, 128), (1, 8, 8, 32), (5, 360, 640, 96), (1, 37, 50, 192),
                                     (2, 33, 64, 152)])
def test_patch_embed(ops, T, H, W, C):
    g = torch.Generator().manual_seed(H)
    x = torch.randn(T, 3, H, W, generator=g)
    w, b = torch.randn(C, 3, 4, 4, generator=g) / 7, torch.randn(C, generator=g)
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    xp = F.pad(x, (0, (4 - W % 4) % 4, 0, (4 - H % 4) % 4))
    ref = F.conv2d(xp, w, b, stride=4).flatten(2).transpose(1, 2)
    ref = F.layer_norm(ref, (C,), ga, be)
    out, Hp, Wp = ops.patch_embed(dev(x), dev(w), dev(b), dev(ga), dev(be))  # MFMA kernel for C in 96/128/192
    close(out.view(T, Hp * Wp, C), ref, 1e-4, 1e-4)
    try:  # exact-fp32 mode keeps the fp32 vector kernels
        ops.set_gemm_mode("f32")
        out32, _, _ = ops.patch_embed(dev(x), dev(w), dev(b), dev(ga), dev(be))
        torch.cuda.synchronize()
    finally:
        ops.set_gemm_mode("f16x3")
    close(out32.view(T, Hp * Wp, C), ref, 1e-4, 1e-4)
    assert (out32 - out).abs().max().item() < 2e-5


@pytest.mark.parametrize("T,H,W,nH,shift", [(2, 18, 25, 3, 0), (2, 18, 25, 3, 3), (1, 9, 13, 6, 3), (1, 7, 7, 1, 3),
                                             (1, 14, 21, 2, 0), (3, 5, 7, 2, 3)])
def test_window_attention(ops, T, H, W, nH, shift):
    g = torch.Generator().manual_seed(H * W + shift)
    C = nH * 32
    x = torch.randn(T, H * W, C, generator=g)  # stands for norm1(x)
    sd = {"attn.qkv.weight": torch.randn(3 * C, C, generator=g) / math.sqrt(C),
          "attn.qkv.bias": torch.randn(3 * C, generator=g) * 0.3,
          "attn.relative_position_bias_table": torch.randn(169, nH, generator=g),
          "attn.proj.weight": torch.eye(C), "attn.proj.bias": torch.zeros(C)}
    ws = 7
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    Hp, Wp = H + pad_b, W + pad_r
    xx = F.pad(x.view(T, H, W, C), (0, 0, 0, pad_r, 0, pad_b))
    if shift:
        xx = torch.roll(xx, shifts=(-shift, -shift), dims=(1, 2))
    xw = xx.view(T, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    am = O.shift_attn_mask(Hp, Wp, ws, ws // 2) if shift else None
    aw = O.window_attention(sd, "attn.", xw, nH, ws, am)
    y = aw.view(T, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(T, Hp, Wp, C)
    if shift:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    ref = y[:, :H, :W].reshape(T * H * W, C)
    qkv = F.linear(x, sd["attn.qkv.weight"], sd["attn.qkv.bias"]).reshape(T * H * W, 3 * C)
    out = ops.window_attn(dev(qkv), dev(sd["attn.qkv.bias"]), dev(sd["attn.relative_position_bias_table"]), T, H, W, C,
                          nH, shift)
    close(out, ref, 1e-4, 1e-4)
    # the VALU form of the same op (kept as the A/B partner of the matrix-core kernel)
    from tce_rvos_amd._lib import lib
    lib().tce_debug_window_attn_set_mfma(0)
    try:
        out2 = ops.window_attn(dev(qkv), dev(sd["attn.qkv.bias"]), dev(sd["attn.relative_position_bias_table"]), T, H, W,
                               C, nH, shift)
    finally:
        lib().tce_debug_window_attn_set_mfma(1)
    close(out2, ref, 1e-4, 1e-4)
    # the exact-fp32 matrix-core kernel (what exact-fp32 mode runs; in the split modes the window goes through the (1,7,7) form of
    # the 3-D split-fp16 kernel): both within fp32 round-off of each other
    lib().tce_debug_window_attn_set_mfma(2)
    try:
        out3 = ops.window_attn(dev(qkv), dev(sd["attn.qkv.bias"]), dev(sd["attn.relative_position_bias_table"]), T, H, W,
                               C, nH, shift)
    finally:
        lib().tce_debug_window_attn_set_mfma(1)
    close(out3, ref, 1e-4, 1e-4)
    assert float((out3 - out).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6
    ops.set_gemm_mode("f32")
    try:
        out4 = ops.window_attn(dev(qkv), dev(sd["attn.qkv.bias"]), dev(sd["attn.relative_position_bias_table"]), T, H, W,
                               C, nH, shift)
    finally:
        ops.set_gemm_mode("f16x3")
    assert torch.equal(out4, out3)


def test_gelu_branch_free_erf_is_fp32_accurate(ops):
    """csrc/common.h tce_erff: the device library's two erf polynomials evaluated branch-free (both + select) with a single
    v_exp_f32: GELU through an identity GEMM (exact products) against torch's fp64 GELU on a dense grid incl. the |x| = 1 seam of
    the two paths (x / sqrt(2) = +-1), large |x| and tiny values."""
    n = 64
    xs = torch.cat([torch.linspace(-8, 8, 40000), torch.linspace(-1.5, -1.3, 4000), torch.linspace(1.3, 1.5, 4000),
                    torch.tensor([0.0, 1e-6, -1e-6, 1e-3, 30.0, -30.0])])
    xs = torch.cat([xs, torch.zeros((-len(xs)) % n)]).view(-1, n)
    eye = torch.eye(n)
    ops.set_gemm_mode("f32")   # exact fp32 products: out = GELU(x) up to the kernel's GELU alone
    try:
        out = ops.gemm_ex(dev(xs), dev(eye), torch.empty(xs.shape[0], n, device="cuda"), xs.shape[0], n, n, n, n, n, act=ops.ACT_GELU).cpu()
    finally:
        ops.set_gemm_mode("f16x3")
    ref = F.gelu(xs.double())
    err = (out.double() - ref).abs()
    # |GELU error| = |x| / 2 * |erf error|: one or two ulps of erf (1.2e-7) at |x| <= 8; as with any fp32 erf-form GELU the far left
    # tail (1 + erf -> 0) is accurate absolutely, not relatively
    assert float(err.max()) < 6e-7, float(err.max())
    core = xs.abs() <= 3
    assert float((err[core] / ref[core].abs().clamp_min(1e-2)).max()) < 2e-5


@pytest.mark.parametrize("M,N,K", [(32, 2304, 768), (32, 768, 3072), (7, 768, 768), (100, 3072, 768), (1, 32, 256)])
def test_thin_linear_weight_stream(ops, M, N, K):
    """csrc/thin.hip: the partial planes of x W^T, finished by tce_splitk_reduce_f32 (bias / GELU / residual / LayerNorm) or
    by the next layer's loads (partial planes + bias + GELU as the x operand), against fp64."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.3
    res = torch.randn(M, N, generator=g)
    splits = ops.thin_splits(M, N, K)
    assert splits == K // 256
    ws = torch.empty(splits, M, N, device="cuda")
    ops.thin_partials(dev(x), dev(w), ws, M, N, K)
    ref = x.double() @ w.double().T
    scale = (x.double().abs() @ w.double().abs().T).max().item()
    assert ((ws.sum(0).cpu().double() - ref).abs().max().item()) < 2e-6 * scale          # fp32-class (3 x fp16 split)
    out = ops.splitk_reduce(ws, splits, M, N, torch.empty(M, N, device="cuda"), bias=dev(b), act=ops.ACT_GELU)
    close(out, F.gelu(F.linear(x, w, b)), 1e-4, 1e-4)
    out = ops.splitk_reduce(ws, splits, M, N, dev(res).clone(), bias=dev(b), res=dev(res), ldres=N, res_mode=ops.RES_ADD)
    close(out, F.linear(x, w, b) + res, 1e-4, 1e-4)
    if N <= 1024:
        ga, be = torch.randn(N, generator=g), torch.randn(N, generator=g)
        buf = dev(res).clone()   # in place on the residual stream
        ops.splitk_reduce(ws, splits, M, N, buf, bias=dev(b), res=buf, ldres=N, res_mode=ops.RES_ADD, ln=(dev(ga), dev(be)), eps=1e-5)
        close(buf, F.layer_norm(F.linear(x, w, b) + res, (N,), ga, be, 1e-5), 2e-4, 2e-4)
    # the planes as the NEXT layer's x operand: y = GELU(x W^T + b) W2^T with no reduction launch in between
    if N % 256 == 0:
        N2 = 64
        w2 = torch.randn(N2, N, generator=g) / math.sqrt(N)
        ws2 = torch.empty(N // 256, M, N2, device="cuda")
        ops.thin_partials(ws, dev(w2), ws2, M, N2, N, xsplits=splits, bias_x=dev(b), act_x=ops.ACT_GELU)
        close(ws2.sum(0), F.linear(F.gelu(F.linear(x, w, b)), w2), 2e-4, 2e-4)
    with ops.arith("f16"):
        ws16 = ops.thin_partials(dev(x), dev(w), torch.empty_like(ws), M, N, K)
    d16 = (ws16.sum(0).cpu().double() - ref).abs().max().item()
    assert 0 < d16 < 2e-3 * scale
    with ops.arith("f32"):
        assert ops.thin_splits(M, N, K) == 0    # exact-fp32 mode keeps the tiled fp32-MFMA GEMM


def _swin_half_ref(x, sd, T, H, W, nH, shift):
    """x + proj(window_attention(norm1(x))) from the oracle's SwinTransformerBlock (pinned to the reference by the e2e
    fixtures): the block with an MLP whose second layer is zero."""
    C = x.shape[-1]
    full = dict(sd)
    full.update({"norm2.weight": torch.ones(C), "norm2.bias": torch.zeros(C), "mlp.fc1.weight": torch.zeros(4 * C, C),
                 "mlp.fc1.bias": torch.zeros(4 * C), "mlp.fc2.weight": torch.zeros(C, 4 * C), "mlp.fc2.bias": torch.zeros(C)})
    Hp, Wp = (H + 6) // 7 * 7, (W + 6) // 7 * 7
    am = O.shift_attn_mask(Hp, Wp, 7, 3) if shift else None
    return O.swin_block(full, "", x.view(T, H * W, C), H, W, nH, 7, shift, am).reshape(T * H * W, C)


@pytest.mark.parametrize("T,H,W,C,shift", [(2, 18, 25, 96, 0), (2, 18, 25, 96, 3),    # ragged in both dims, two frames
                                            (1, 9, 13, 192, 3), (1, 7, 7, 128, 3),      # one window per dim: every mask region
                                            (1, 14, 21, 256, 0), (3, 5, 7, 128, 3),     # fewer rows than a window
                                            (1, 23, 40, 256, 3), (3, 45, 80, 192, 3),   # config 2's stage-1 / stage-2 maps
                                            (1, 13, 9, 96, 3)])                         # an odd number of windows (idle window slot)
def test_swin_attn_fused(ops, T, H, W, C, shift):
    """The one-launch Swin attention half-block (csrc/swinattn.hip) against the oracle's SwinTransformerBlock: norm1, padding
    after the norm, cyclic shift, windows, qkv, bias table, -100 mask, softmax, AV, proj, residual -- shifted / padded /
    ragged cases, in place and out of place (bit-identical), the single-pass fp16 mode's error class."""
    g = torch.Generator().manual_seed(H * W + C + shift)
    nH = C // 32
    x = torch.randn(T * H * W, C, generator=g)
    sd = {"norm1.weight": 1 + 0.1 * torch.randn(C, generator=g), "norm1.bias": 0.1 * torch.randn(C, generator=g),
          "attn.qkv.weight": torch.randn(3 * C, C, generator=g) / math.sqrt(C), "attn.qkv.bias": torch.randn(3 * C, generator=g) * 0.3,
          "attn.relative_position_bias_table": torch.randn(169, nH, generator=g),
          "attn.proj.weight": torch.randn(C, C, generator=g) / math.sqrt(C), "attn.proj.bias": torch.randn(C, generator=g) * 0.2}
    ref = _swin_half_ref(x, sd, T, H, W, nH, shift)
    d = {k: dev(v) for k, v in sd.items()}
    pk = ops.swin_attn_pack(d["attn.qkv.weight"], d["attn.proj.weight"])
    args = (pk, d["attn.qkv.bias"], d["attn.proj.bias"], d["attn.relative_position_bias_table"], d["norm1.weight"], d["norm1.bias"],
            T, H, W, C, shift)
    xd = dev(x)
    out = torch.full_like(xd, float("nan"))
    ops.swin_attn_fused(xd, *args, out=out)
    close(out, ref, 1e-4, 2e-4)
    assert torch.equal(xd.cpu(), x)                      # out of place: x untouched
    xin = xd.clone()
    ops.swin_attn_fused(xin, *args)                      # in place on the residual stream
    assert torch.equal(xin, out)
    rel = ((out.cpu().double() - ref.double()).abs().max() / ref.abs().max()).item()
    assert rel < 2e-5, rel                               # fp32-class (3 x fp16 split)
    with ops.arith("f16"):
        pk16 = ops.swin_attn_pack(d["attn.qkv.weight"], d["attn.proj.weight"])
        out16 = ops.swin_attn_fused(xd, pk16, *args[1:], out=torch.empty_like(xd))
    d16 = (out16.cpu() - ref).abs().max().item()
    assert 0 < d16 < 5e-2, d16                           # one fp16 MFMA per product: fp16-class error, not garbage
    ops.check_range()


def test_swin_attn_fused_matches_three_launch_form_at_config2_size(ops):
    """BASELINE config 2's first Swin stage (5 x 90 x 160 tokens, C = 96, shifted block): the fused launch against the
    three-launch form it replaces (LayerNorm -> qkv GEMM, window attention kernel, proj GEMM + residual), every row."""
    T, H, W, C, shift = 5, 90, 160, 96, 3
    g = torch.Generator().manual_seed(5)
    x = dev(torch.randn(T * H * W, C, generator=g))
    wqkv, bqkv = dev(torch.randn(3 * C, C, generator=g) / math.sqrt(C)), dev(torch.randn(3 * C, generator=g) * 0.3)
    wp, bp = dev(torch.randn(C, C, generator=g) / math.sqrt(C)), dev(torch.randn(C, generator=g) * 0.2)
    table, g1, b1 = dev(torch.randn(169, 3, generator=g)), dev(1 + 0.1 * torch.randn(C, generator=g)), dev(0.1 * torch.randn(C, generator=g))
    xn = ops.layernorm(x, g1, b1)
    qkv = ops.gemm(xn, wqkv, bias=bqkv)
    att = ops.window_attn(qkv, bqkv, table, T, H, W, C, 3, shift)
    ref = ops.gemm(att, wp, bias=bp, res=x, res_mode=ops.RES_ADD)
    out = ops.swin_attn_fused(x, ops.swin_attn_pack(wqkv, wp), bqkv, bp, table, g1, b1, T, H, W, C, shift, out=torch.empty_like(x))
    torch.cuda.synchronize()
    d = (out - ref).abs().max().item()
    print("fused vs three launches: max abs diff", d)
    assert d < 2e-4


@pytest.mark.parametrize("T,H,W,C", [(2, 18, 25, 96), (1, 9, 13, 192), (1, 5, 7, 384), (1, 4, 4, 32)])
def test_patch_merge_ln(ops, T, H, W, C):
    g = torch.Generator().manual_seed(H + C)
    x = torch.randn(T, H, W, C, generator=g)
    ga, be = torch.randn(4 * C, generator=g), torch.randn(4 * C, generator=g)
    xp = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    cat = torch.cat([xp[:, 0::2, 0::2], xp[:, 1::2, 0::2], xp[:, 0::2, 1::2], xp[:, 1::2, 1::2]], -1)
    ref = F.layer_norm(cat.reshape(-1, 4 * C), (4 * C,), ga, be)
    out, H2, W2 = ops.patch_merge_ln(dev(x.reshape(-1, C)), dev(ga), dev(be), T, H, W, C)
    close(out, ref, 1e-4, 1e-5)


@pytest.mark.parametrize("batch,nh,Lq,Lk,masked", [(1, 8, 1200, 1200, False), (5, 8, 700, 8, False), (1, 8, 40, 40, False),
                                                    (5, 2, 5, 5, False), (1, 8, 300, 32, True), (2, 4, 65, 33, True),
                                                    # Lk >= 256: the split-fp16 kernel (one wave / four waves per workgroup)
                                                    (1, 8, 2300, 2300, False), (5, 8, 3600, 301, True), (2, 3, 77, 257, True),
                                                    # few query tiles, >= 1024 keys: keys split over the waves of a workgroup
                                                    (1, 8, 4600, 4600, False), (2, 4, 100, 1101, True), (1, 2, 33, 1024, False)])
def test_mha_core(ops, batch, nh, Lq, Lk, masked):
    g = torch.Generator().manual_seed(Lq + Lk)
    E = nh * 32
    q, k, v = (torch.randn(batch, L, E, generator=g) for L in (Lq, Lk, Lk))
    km = None
    if masked:
        km = torch.rand(batch, Lk, generator=g) < 0.3
        km[:, 0] = False
    qh = q.view(batch, Lq, nh, 32).transpose(1, 2) * (32 ** -0.5)
    kh, vh = k.view(batch, Lk, nh, 32).transpose(1, 2), v.view(batch, Lk, nh, 32).transpose(1, 2)
    att = qh @ kh.transpose(-1, -2)
    if masked:
        att = att.masked_fill(km[:, None, None, :], float("-inf"))
    ref = (torch.softmax(att, -1) @ vh).transpose(1, 2).reshape(batch, Lq, E)
    out = torch.empty(batch, Lq, E, device="cuda")
    ops.mha_core(dev(q), dev(k), dev(v), batch, nh, Lq, Lk, E, E, E, Lq * E, Lk * E, Lk * E, out, E, Lq * E,
                 kmask=dev(km.to(torch.uint8)) if masked else None)
    close(out, ref, 1e-4, 1e-5)
    if Lk >= 256:  # exact-fp32 mode runs the fp32-MFMA kernel: both must agree to fp32 round-off
        exact = torch.empty_like(out)
        try:
            ops.set_gemm_mode("f32")
            ops.mha_core(dev(q), dev(k), dev(v), batch, nh, Lq, Lk, E, E, E, Lq * E, Lk * E, Lk * E, exact, E, Lq * E,
                         kmask=dev(km.to(torch.uint8)) if masked else None)
            torch.cuda.synchronize()
        finally:
            ops.set_gemm_mode("f16x3")
        close(exact, ref, 1e-4, 1e-5)
        assert (exact - out).abs().max().item() < 5e-6
    if Lk >= 1024:  # pre-split form (tce_mha_ws_f32): K / V planes from a caller-provided workspace
        ws_out = torch.empty_like(out)
        ops.mha_core(dev(q), dev(k), dev(v), batch, nh, Lq, Lk, E, E, E, Lq * E, Lk * E, Lk * E, ws_out, E, Lq * E,
                     kmask=dev(km.to(torch.uint8)) if masked else None,
                     alloc=lambda n: torch.empty(n, dtype=torch.float32, device="cuda"))
        close(ws_out, ref, 1e-4, 1e-5)
        assert (ws_out - out).abs().max().item() < 5e-6


def test_msda_reference_op_matches_reference_fixture(ops):
    """The drop-in for ms_deform_attn_forward against outputs of the reference's own core (golden fixture), INCLUDING the
    reference's own known-answer case (models/ops/test.py:21-26: D = 2); tolerance is the reference's own float check
    (models/ops/test.py:56: rtol 1e-2, atol 1e-3) -- we ask for 1e-4."""
    fx = load_npz("msda_cases.npz")
    dims = set()
    for i in range(int(fx["n_cases"])):
        shapes = torch.from_numpy(fx[f"c{i}_shapes"])
        value = torch.from_numpy(fx[f"c{i}_value"])
        dims.add(value.shape[-1])
        lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
        out = ops.ms_deform_attn_forward(dev(value), shapes.cuda(), lsi.cuda(), dev(torch.from_numpy(fx[f"c{i}_loc"])),
                                         dev(torch.from_numpy(fx[f"c{i}_w"])))
        close(out, torch.from_numpy(fx[f"c{i}_out"]), 1e-4, 1e-5)
    assert 2 in dims and 32 in dims


def test_msda_backward_matches_reference_fixture(ops):
    """tce_ms_deform_attn_backward_f32 against gradients of the reference's own core (golden fixture; D = 2 and D = 32,
    out-of-range samples); grad_value sums atomically, hence the small absolute slack."""
    fx = load_npz("msda_cases.npz")
    for i in range(int(fx["n_cases"])):
        shapes = torch.from_numpy(fx[f"c{i}_shapes"])
        lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
        t = lambda k: dev(torch.from_numpy(fx[f"c{i}_{k}"]))
        gv, gl, gw = ops.ms_deform_attn_backward(t("value"), shapes.cuda(), lsi.cuda(), t("loc"), t("w"), t("gout"))
        for got, key in ((gv, "gvalue"), (gl, "gloc"), (gw, "gw")):
            ref = torch.from_numpy(fx[f"c{i}_{key}"])
            close(got, ref, 1e-4, 1e-5 * max(1.0, ref.abs().max().item()))


@pytest.mark.parametrize("N,Lq,M,Dh,L,P", [(2, 17, 3, 30, 2, 3), (1, 9, 2, 71, 3, 2), (2, 5, 8, 64, 4, 4), (1, 300, 8, 32, 4, 4)])
def test_msda_autograd_function_generic_shapes(ops, N, Lq, M, Dh, L, P):
    """The autograd wrapper (the reference's MSDeformAttnFunction interface) on head dims of the reference's own gradient
    test (models/ops/test.py:85-86) against the oracle's backward; many queries colliding on few value rows."""
    g = torch.Generator().manual_seed(7 * N + Dh + L * P)
    shapes = torch.tensor([(6 + 2 * l, 5 + 3 * l) for l in range(L)][::-1], dtype=torch.int64)
    S = int(shapes.prod(1).sum())
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    value = torch.randn(N, S, M, Dh, generator=g)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.6 - 0.3
    aw = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    go = torch.randn(N, Lq, M * Dh, generator=g)
    v, p, w = (dev(x).requires_grad_(True) for x in (value, loc, aw))
    out = ops.MSDeformAttnFunction.apply(v, shapes.cuda(), lsi.cuda(), p, w, 64)
    out.backward(dev(go))
    rv, rp, rw = O.msda_core_backward(value.double(), [(int(h), int(ww)) for h, ww in shapes], loc.double(), aw.double(),
                                      go.double())
    close(out.detach(), O.msda_core(value, [(int(h), int(ww)) for h, ww in shapes], loc, aw), 1e-4, 1e-5)
    for got, ref in ((v.grad, rv), (p.grad, rp), (w.grad, rw)):
        close(got, ref.float(), 1e-4, 2e-5 * max(1.0, ref.abs().max().item()))


@pytest.mark.parametrize("N,Lq,M,Dh,L,P", [(2, 50, 8, 32, 4, 8),     # L*P = 32: dword row-gather kernel
                                           (1, 33, 4, 32, 8, 8),     # L*P = 64: generic kernel
                                           (2, 17, 3, 30, 2, 3), (1, 9, 2, 71, 3, 2), (2, 5, 8, 64, 4, 4),  # test.py:85-86 dims
                                           (1, 40, 8, 32, 4, 4)])    # the 16-byte gather kernel
def test_msda_reference_op_generic_shapes(ops, N, Lq, M, Dh, L, P):
    """Head dims and point counts beyond the model's (8 x 32, 4 x 4) against the oracle's restatement of the reference
    core (itself pinned to the reference by the fixture test above); locations include out-of-range samples."""
    g = torch.Generator().manual_seed(N * 1000 + Dh + L * P)
    shapes = torch.tensor([(6 + 2 * l, 5 + 3 * l) for l in range(L)][::-1], dtype=torch.int64)
    S = int(shapes.prod(1).sum())
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    value = torch.randn(N, S, M, Dh, generator=g)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.6 - 0.3
    aw = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    out = ops.ms_deform_attn_forward(dev(value), shapes.cuda(), lsi.cuda(), dev(loc), dev(aw))
    ref = O.msda_core(value, [(int(h), int(w)) for h, w in shapes], loc, aw)
    close(out, ref, 1e-4, 1e-5)


@pytest.mark.parametrize("N,Lq,ref_dim", [(2, 300, 2), (3, 5, 4), (1, 8, 2),   # <= 8192 items: one wave per item (few-query form)
                                          (2, 700, 2), (1, 1500, 4)])          # more: 8 lanes per item
def test_msda_fused(ops, N, Lq, ref_dim):
    g = torch.Generator().manual_seed(Lq)
    M, L, P = 8, 4, 4
    shapes = [(9, 13), (5, 7), (3, 4), (2, 2)]
    S = sum(h * w for h, w in shapes)
    value = torch.randn(N, S, M, 32, generator=g)
    proj = torch.randn(N, Lq, M * L * P * 3, generator=g)
    proj[..., :M * L * P * 2] *= 2.0
    ref = torch.rand(N, Lq, ref_dim, generator=g) * 1.2 - 0.1
    off = proj[..., :M * L * P * 2].view(N, Lq, M, L, P, 2)
    aw = torch.softmax(proj[..., M * L * P * 2:].view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    refl = ref[:, :, None, :].expand(N, Lq, L, ref_dim)
    if ref_dim == 2:
        norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
        loc = refl[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = refl[:, :, None, :, None, :2] + off / P * refl[:, :, None, :, None, 2:] * 0.5
    expect = O.msda_core(value, shapes, loc, aw)
    out = ops.msda_fused(dev(value), dev(proj), dev(ref), shapes, N, S, M, Lq, L, P, ref_dim, True)
    close(out.view(N, Lq, M * 32), expect, 1e-4, 1e-4)


@pytest.mark.parametrize("N,Lq,ref_dim,padded", [(5, 8, 2, False), (3, 5, 4, False), (2, 8, 2, True), (40, 8, 2, False), (1, 1, 4, True)])
def test_msda_fewq_raw_sample_then_project(ops, N, Lq, ref_dim, padded):
    """tce_msda_fewq_raw_f32 ("sample, then project": the few-query attention core on the UN-projected rows, value_proj applied to
    the bilinear samples) against the module's own order -- value = value_proj(src), padded rows zero-filled, then the gather
    (ops/modules/ms_deform_attn.py:95-114) -- on the oracle, with out-of-range locations (missing corners drop their share of the
    bias too) and padded levels."""
    g = torch.Generator().manual_seed(N * 100 + Lq + ref_dim)
    M, L, P = 8, 4, 4
    shapes = [(9, 13), (5, 7), (3, 4), (2, 2)]
    valid = [(7, 10), (4, 5), (2, 3), (1, 2)] if padded else None
    S = sum(h * w for h, w in shapes)
    src = torch.randn(N, S, 256, generator=g)
    wv, bv = torch.randn(256, 256, generator=g) / 16, torch.randn(256, generator=g)
    proj = torch.randn(N, Lq, M * L * P * 3, generator=g)
    proj[..., :M * L * P * 2] *= 2.0
    ref = torch.rand(N, Lq, ref_dim, generator=g) * 1.2 - 0.1
    off = proj[..., :M * L * P * 2].view(N, Lq, M, L, P, 2)
    aw = torch.softmax(proj[..., M * L * P * 2:].view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    vr = torch.tensor([[(wv_ / w), (hv_ / h)] for (h, w), (hv_, wv_) in zip(shapes, valid or shapes)], dtype=torch.float32)
    if ref_dim == 2:
        refl = ref[:, :, None, :] * vr[None, None]
        norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
        loc = refl[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        refl = ref[:, :, None, :] * torch.cat([vr, vr], -1)[None, None]
        loc = refl[:, :, None, :, None, :2] + off / P * refl[:, :, None, :, None, 2:] * 0.5
    value = F.linear(src.double(), wv.double(), bv.double()).float().view(N, S, M, 32)
    if padded:
        pad = torch.cat([torch.ones(h, w, dtype=torch.bool).index_put_(
            (torch.arange(hv_)[:, None], torch.arange(wv_)[None, :]), torch.tensor(False)).reshape(-1)
            for (h, w), (hv_, wv_) in zip(shapes, valid)])
        value = value.masked_fill(pad[None, :, None, None], 0.0)
    expect = O.msda_core(value, shapes, loc, aw)
    out = ops.msda_fewq_raw(dev(src.view(N * S, 256)), dev(wv), dev(bv), dev(proj), dev(ref), shapes, N, S, Lq, L, P, ref_dim, True,
                            valid_hw=valid)
    close(out.view(N, Lq, 256), expect, 1e-4, 1e-4)
    # and against the projected few-query kernel on the HIP-projected value (the path it replaces)
    vd = ops.gemm(dev(src.view(N * S, 256)), dev(wv), bias=dev(bv))
    old = ops.msda_fused(vd, dev(proj), dev(ref), shapes, N, S, M, Lq, L, P, ref_dim, True, valid_hw=valid)
    assert float((old - out).abs().max()) <= 2e-5 * float(expect.abs().max()) + 1e-6


@pytest.mark.parametrize("T,S,fpc", [(3, 168, 3), (5, 4820, 5), (8, 1000, 2), (1, 7, 1)])
def test_contrastive_cosine(ops, T, S, fpc):
    """tce_contrastive_f32 (contrastive_cal, tce_rvos.py:512-521): cosine similarity between the mean of a frame's memory rows and
    its clip's sentence feature, against torch's F.cosine_similarity in fp64 -- several clips per launch (frames_per_clip), row
    counts that do not divide into the 32 chunks, a near-zero sentence feature (the eps clamp)."""
    g = torch.Generator().manual_seed(T * S)
    mem = torch.randn(T, S, 256, generator=g) + 0.3
    sent = torch.randn(T // fpc, 256, generator=g)
    if T == 8:
        sent[1] *= 1e-9   # |x| |y| below eps: the clamp decides
    out, ws = torch.empty(T, device="cuda"), torch.empty(T * 32 * 256, device="cuda")
    ops.contrastive(dev(mem), dev(sent), T, S, 256, fpc, out, ws)
    ref = F.cosine_similarity(mem.double().mean(1), sent.double().repeat_interleave(fpc, 0), dim=1, eps=1e-6)
    close(out, ref.float(), 1e-5, 1e-6)


def test_pos_sine2d(ops):
    for T, h, w in [(2, 9, 13), (1, 45, 80)]:
        ref = O.pos_sine_2d(torch.zeros(T, h, w, dtype=torch.bool), 128).permute(0, 2, 3, 1).reshape(T * h * w, 256)
        close(ops.pos_sine2d(T, h, w, 128, "cuda"), ref, 1e-4, 2e-5)
        add = torch.randn(256)
        close(ops.pos_sine2d(T, h, w, 128, "cuda", add=dev(add)), ref + add, 1e-4, 2e-5)


def test_resize(ops):
    fx = load_npz("interp_cases.npz")
    for i in range(int(fx["n_pairs"])):
        x = torch.from_numpy(fx[f"p{i}_in"])  # [2,3,h,w] -> treat (2) as T and pad channels to 4
        T, c, h, w = x.shape
        ho, wo = (int(v) for v in fx[f"p{i}_size"])
        x4 = torch.cat([x, x[:, :1]], 1)
        cl = x4.permute(0, 2, 3, 1).reshape(T * h * w, 4)
        outn = ops.resize_nearest(dev(cl), T, h, w, ho, wo, 4).view(T, ho, wo, 4).permute(0, 3, 1, 2)[:, :3]
        assert torch.equal(outn.cpu(), torch.from_numpy(fx[f"p{i}_nearest"])), i
        outb = ops.resize_bilinear(dev(cl), T, h, w, ho, wo, 4).view(T, ho, wo, 4).permute(0, 3, 1, 2)[:, :3]
        close(outb, torch.from_numpy(fx[f"p{i}_bilinear"]), 1e-5, 1e-5)
        add = torch.randn(T * ho * wo, 4)
        plain = ops.resize_bilinear(dev(cl), T, h, w, ho, wo, 4).cpu()
        close(ops.resize_bilinear(dev(cl), T, h, w, ho, wo, 4, add=dev(add)), plain + add, 1e-6, 1e-6)
        plain_n = ops.resize_nearest(dev(cl), T, h, w, ho, wo, 4).cpu()
        close(ops.resize_nearest(dev(cl), T, h, w, ho, wo, 4, add=dev(add)), plain_n + add, 1e-6, 1e-6)


def test_small_elementwise(ops):
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(25, 256, generator=g), torch.randn(5, 256, generator=g)
    close(ops.add(dev(a), dev(b)), (a.view(5, 5, 256) + b[None]).view(25, 256), 0, 0)
    close(ops.sigmoid(dev(a)), torch.sigmoid(a), 1e-6, 1e-6)
    tmp = torch.randn(25, 4, generator=g)
    for rd in (2, 4):
        ref = torch.rand(25, rd, generator=g)
        ref[0, 0], ref[1, 1] = 0.0, 1.0
        t2 = tmp.clone()
        t2[:, :rd] += O.inverse_sigmoid(ref)
        close(ops.box_refine(dev(tmp), dev(ref)), torch.sigmoid(t2), 1e-5, 1e-6)


@pytest.mark.parametrize("nl,T,Q,h,w", [(3, 2, 5, 18, 25), (3, 2, 30, 9, 15), (1, 1, 1, 5, 131)])
def test_dynamic_mask_head(ops, nl, T, Q, h, w):
    """(3, 2, 30): 90 (level, query) items = two item chunks of the pixel-stationary kernel; (1, 1, 1): a single item."""
    g = torch.Generator().manual_seed(9)
    Cm = 64
    cfg = O.OracleConfig(mask_dim=Cm)
    npar = 8 * (Cm + 2) + 64 + 8 + 8 + 8 + 1
    feats = torch.randn(T, Cm, h, w, generator=g)
    params = torch.randn(nl, T * Q, npar, generator=g) * 0.2
    refs = torch.rand(nl, T * Q, 4, generator=g)
    img = (h * 4 - 2, w * 4 - 1)
    expect = torch.stack([O.dynamic_mask_head(cfg, feats, params[l], refs[l, :, :2], img) for l in range(nl)])
    feats_cl = dev(feats.permute(0, 2, 3, 1).reshape(T, h * w, Cm))
    w0f = torch.empty(T, nl * Q * 8, Cm, device="cuda")
    tail = torch.empty(nl, T * Q, 112, device="cuda")
    ops.mask_pack(dev(params), nl, T, Q, Cm, w0f, tail)
    G = torch.empty(T, h * w, nl * Q * 8, device="cuda")
    ops.gemm_batched(feats_cl, w0f, G)
    masks = torch.empty(nl, T, Q, h, w, device="cuda")
    ops.mask_tail(G, tail, dev(refs), 4, masks, nl, T, Q, h, w, img[0], img[1])
    close(masks.view(nl, T * Q, h, w), expect, 1e-4, 1e-4)


@pytest.mark.parametrize("T,H,W,nH,shifted", [(3, 18, 25, 3, False), (3, 18, 25, 3, True), (9, 9, 13, 2, True),
                                               (9, 3, 4, 2, True), (8, 7, 7, 1, True), (17, 8, 6, 1, True),
                                               # full (8,7,7) windows = 392 keys = 13 key tiles (12 whole + 8 keys), ragged edges
                                               (8, 16, 23, 2, False), (8, 16, 23, 2, True), (16, 14, 7, 1, True)])
def test_window_attention_3d(ops, T, H, W, nH, shifted):
    """3-D window attention core against the oracle's Video-Swin block internals (which are pinned to the
    reference by e2e_vswin_t_small.npz): the matrix-core kernel (3 x fp16 split), its single-pass fp16 mode (error
    class only) and the VALU kernel (exact fp32; the A/B partner and the exact-fp32 mode's kernel)."""
    g = torch.Generator().manual_seed(T * H + W)
    C = nH * 32
    x = torch.randn(1, T, H, W, C, generator=g)  # stands for norm1(x)
    sd = {"attn.qkv.weight": torch.randn(3 * C, C, generator=g) / math.sqrt(C),
          "attn.qkv.bias": torch.randn(3 * C, generator=g) * 0.3,
          "attn.relative_position_bias_table": torch.randn(15 * 13 * 13, nH, generator=g),
          "attn.proj.weight": torch.eye(C), "attn.proj.bias": torch.zeros(C)}
    full = (8, 7, 7)
    ws, ss = O.get_window_size_3d((T, H, W), full, tuple(i // 2 for i in full) if shifted else (0, 0, 0))
    pd, pb, pr = (ws[0] - T % ws[0]) % ws[0], (ws[1] - H % ws[1]) % ws[1], (ws[2] - W % ws[2]) % ws[2]
    xx = F.pad(x, (0, 0, 0, pr, 0, pb, 0, pd))
    Dp, Hp, Wp = T + pd, H + pb, W + pr
    am = None
    if any(i > 0 for i in ss):
        xx = torch.roll(xx, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
        am = O.compute_mask_3d(Dp, Hp, Wp, ws, ss)
    xw = O.window_partition_3d(xx, ws)
    aw = O.window_attention_3d(sd, "attn.", xw, nH, full, am)
    y = O.window_reverse_3d(aw, ws, 1, Dp, Hp, Wp)
    if any(i > 0 for i in ss):
        y = torch.roll(y, shifts=(ss[0], ss[1], ss[2]), dims=(1, 2, 3))
    ref = y[:, :T, :H, :W].reshape(T * H * W, C)
    qkv = F.linear(x, sd["attn.qkv.weight"], sd["attn.qkv.bias"]).reshape(T * H * W, 3 * C)
    args = (dev(qkv), dev(sd["attn.qkv.bias"]), dev(sd["attn.relative_position_bias_table"]), T, H, W, C, nH, shifted)
    out = ops.window_attn3d(*args)
    close(out, ref, 1e-4, 1e-4)
    from tce_rvos_amd._lib import lib
    lib().tce_debug_window_attn_set_mfma(0)
    try:
        out2 = ops.window_attn3d(*args)
    finally:
        lib().tce_debug_window_attn_set_mfma(1)
    close(out2, ref, 1e-4, 1e-4)
    assert (out - out2).abs().max().item() < 2e-5   # the split keeps the matrix-core kernel fp32-accurate
    with ops.arith("f16"):
        out3 = ops.window_attn3d(*args)
    d3 = (out3.cpu() - ref).abs().max().item()
    assert 0 < d3 < 2e-2, d3   # one fp16 MFMA per product: fp16-class error, not garbage
    with ops.arith("f32"):     # exact-fp32 mode runs the VALU kernel: bit-identical to the A/B partner
        assert torch.equal(ops.window_attn3d(*args), out2)


def test_harness_select_masks_matches_reference_caller(ops):
    """tce_select_masks_u8 against the reference caller's outputs (golden harness_cases.npz)."""
    fx = load_npz("harness_cases.npz")
    for i in range(int(fx["n_cases"])):
        size = tuple(int(v) for v in fx[f"h{i}_size"])
        m, best = ops.select_masks(dev(torch.from_numpy(fx[f"h{i}_logits"])[0]), dev(torch.from_numpy(fx[f"h{i}_masks"])[0]), size)
        assert int(best.item()) == int(fx[f"h{i}_best"])
        ref = torch.from_numpy(fx[f"h{i}_out"])
        assert O.mask_iou(m.cpu().bool(), ref) > 1 - 1e-4


@pytest.mark.parametrize("T,H,W", [(1, 72, 100), (2, 37, 61), (1, 360, 640)])
def test_resnet_stem_and_maxpool(ops, T, H, W):
    """conv 7x7/s2/p3 + folded frozen BN + ReLU, then MaxPool2d(3, 2, 1), vs PyTorch on the CPU (oracle ops)."""
    g = torch.Generator().manual_seed(H)
    x = torch.randn(T, 3, H, W, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) / 12.0
    b = torch.randn(64, generator=g) * 0.1
    ref = F.relu(F.conv2d(x, wt, b, stride=2, padding=3))
    refp = F.max_pool2d(ref, 3, stride=2, padding=1)
    out, Ho, Wo = ops.resnet_stem(x.cuda(), wt.reshape(64, 147).t().contiguous().cuda(), b.cuda())
    assert (Ho, Wo) == tuple(ref.shape[-2:])
    got = out.view(T, Ho, Wo, 64).permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() < 2e-5
    pooled, Hp, Wp = ops.maxpool3x3s2_cl(out, T, Ho, Wo, 64)
    assert (Hp, Wp) == tuple(refp.shape[-2:])
    gotp = pooled.view(T, Hp, Wp, 64).permute(0, 3, 1, 2).cpu()
    assert torch.equal(gotp, F.max_pool2d(got, 3, stride=2, padding=1))  # the pool itself is exact
    assert (gotp - refp).abs().max().item() < 2e-5


@pytest.mark.parametrize("stride,k,cin,cout", [(1, 1, 64, 256), (2, 1, 256, 512), (2, 3, 128, 128), (1, 3, 64, 64)])
def test_conv_with_identity_and_relu_after_residual(ops, stride, k, cin, cout):
    """ResNet bottleneck epilogue: relu(conv(x) + bias + identity) (act 3, res_mode 1), strided 1x1 / 3x3."""
    g = torch.Generator().manual_seed(cin + k)
    T, H, W = 2, 23, 31
    x = torch.randn(T, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    y = F.conv2d(x, wt, b, stride=stride, padding=k // 2)
    idt = torch.randn(y.shape, generator=g)
    ref = F.relu(y + idt)
    xcl = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().cuda()
    wcl = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().cuda()
    icl = idt.permute(0, 2, 3, 1).reshape(-1, cout).contiguous().cuda()
    out, Ho, Wo = ops.conv2d_cl(xcl, wcl, T, H, W, cin, k, k, stride, k // 2, bias=b.cuda(), act=ops.ACT_RELU_AFTER_RES,
                                res=icl, res_mode=ops.RES_ADD)
    got = out.view(T, Ho, Wo, cout).permute(0, 3, 1, 2).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("M,N,K,splits", [(32, 768, 3072, 16), (32, 2304, 768, 4), (25, 256, 2048, 8), (1, 768, 768, 2),
                                          (100, 2152, 512, 4)])
@pytest.mark.parametrize("act,res_mode", [(0, 0), (2, 0), (0, 1), (3, 1)])
def test_gemm_splitk(ops, M, N, K, splits, act, res_mode):
    """Split-K path (skinny deep GEMMs: RoBERTa at 32 tokens, decoder FFNs) == the plain GEMM epilogue semantics."""
    g = torch.Generator().manual_seed(M + N + K)
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = F.linear(a, w, b)
    if act == 2:
        ref = F.gelu(ref)
    if res_mode == 1:
        ref = ref + r
    if act == 3:
        ref = F.relu(ref)
    out = torch.empty(M, N, device="cuda")
    ws = torch.empty(splits * M * N, device="cuda")
    ops.gemm_ex(dev(a), dev(w), out, M, N, K, K, K, N, bias=dev(b), act=act, res=dev(r) if res_mode else None, ldres=N,
                res_mode=res_mode, splitk=splits, ws=ws)
    close(out, ref, 1e-4, 2e-4)
    was, ops.SPLITK_ENABLED = ops.SPLITK_ENABLED, True
    try:
        assert ops.splitk_for(32, 768, 3072) == 16 and ops.splitk_for(24100, 256, 2048) == 1 and ops.splitk_for(32, 768, 768) == 4
    finally:
        ops.SPLITK_ENABLED = was


@pytest.mark.parametrize("act,res_mode", [(1, 2), (2, 1), (2, 2), (3, 0), (3, 2), (1, 1)])
def test_gemm_unspecialised_epilogue_combinations(ops, act, res_mode):
    """Epilogue combinations without a specialised kernel body run as GEMM + one elementwise pass: same semantics
    (activation, then residual, act 3 = ReLU after the residual)."""
    g = torch.Generator().manual_seed(act * 3 + res_mode)
    M, N, K = 300, 200, 96
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = F.linear(a, w, b)
    if act == 1:
        ref = F.relu(ref)
    if act == 2:
        ref = F.gelu(ref)
    if res_mode == 1:
        ref = ref + r
    if res_mode == 2:
        ref = ref * r
    if act == 3:
        ref = F.relu(ref)
    out = ops.gemm(dev(a), dev(w), bias=dev(b), act=act, res=dev(r) if res_mode else None, res_mode=res_mode)
    close(out, ref, 1e-4, 2e-4)


@pytest.mark.parametrize("T,H,W,cin,cout,stride,splits", [(5, 12, 20, 768, 256, 2, 8), (1, 9, 7, 64, 96, 1, 3), (2, 6, 10, 2048, 256, 2, 6)])
def test_conv_splitk(ops, T, H, W, cin, cout, stride, splits):
    """Split-K implicit-GEMM convolution (the level-3 3x3/s2 conv of C5: few output rows, K = 9*C5)."""
    g = torch.Generator().manual_seed(cin + splits)
    x = torch.randn(T, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x, wt, b, stride=stride, padding=1)
    xcl = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().cuda()
    wcl = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().cuda()
    Ho, Wo = ref.shape[-2:]
    ws = torch.empty(splits * T * Ho * Wo * cout, device="cuda")
    out, ho, wo = ops.conv2d_cl(xcl, wcl, T, H, W, cin, 3, 3, stride, 1, bias=b.cuda(), splitk=splits, ws=ws)
    assert (ho, wo) == (Ho, Wo)
    got = out.view(T, Ho, Wo, cout).permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() < 3e-4


@pytest.mark.parametrize("M,C,Hd,act,ln", [
    (24100, 256, 2048, "relu", "out"),   # encoder / FTF FFN (tce_deformable_transformer.py:489-491,548-552)
    (18000, 256, 2048, "relu", "out"),   # VisionLanguageBlock FFN, stride 8 (segmentation.py:374-376)
    (72000, 256, 2048, "relu", "out"),   # stride 4
    (72000, 96, 384, "gelu", "in"),      # Swin-T stage-1 MLP (swin_transformer.py:28-47,255-256)
    (4600, 192, 768, "gelu", "in"),
    (1201, 128, 512, "gelu", "in"),      # Swin-B widths, ragged row count
    (77, 256, 1024, "gelu", "in"),
    (333, 256, 64, "relu", "both"),
    (129, 96, 96, "relu", "none"),
])
def test_ffn_fused(ops, M, C, Hd, act, ln):
    """The fused FFN / MLP launch (hidden tensor on chip) against torch fp32 of the op sequence it replaces; the
    accuracy class is checked against fp64 on a slice of rows."""
    g = torch.Generator().manual_seed(M + C + Hd)
    x = torch.randn(M, C, generator=g)
    w1 = torch.randn(Hd, C, generator=g) / math.sqrt(C)
    b1 = torch.randn(Hd, generator=g) * 0.2
    w2 = torch.randn(C, Hd, generator=g) / math.sqrt(Hd)
    b2 = torch.randn(C, generator=g) * 0.2
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    gam2, bet2 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2

    def ref(x_, dt):
        y = F.layer_norm(x_, (C,), gam.to(dt), bet.to(dt), 1e-5) if ln in ("in", "both") else x_
        h = F.linear(y, w1.to(dt), b1.to(dt))
        h = torch.relu(h) if act == "relu" else F.gelu(h)
        o = x_ + F.linear(h, w2.to(dt), b2.to(dt))
        return F.layer_norm(o, (C,), gam2.to(dt), bet2.to(dt), 1e-5) if ln in ("out", "both") else o

    pk = ops.ffn_pack(dev(w1), dev(b1), dev(w2))
    xd = dev(x)
    out = torch.empty_like(xd)
    ops.ffn_fused(xd, pk, dev(b2), Hd, ops.ACT_RELU if act == "relu" else ops.ACT_GELU,
                  ln_in=(dev(gam), dev(bet)) if ln in ("in", "both") else None,
                  ln_out=(dev(gam2), dev(bet2)) if ln in ("out", "both") else None, out=out)
    rows = torch.cat([torch.arange(0, min(M, 300)), torch.arange(max(0, M - 200), M)]).unique()
    r32 = ref(x[rows], torch.float32)
    close(out[rows.cuda()], r32, 2e-4, 2e-4)
    r64 = ref(x[rows].double(), torch.float64)
    e_kernel = (out[rows.cuda()].cpu().double() - r64).abs().max().item()
    e_torch = (r32.double() - r64).abs().max().item()
    assert e_kernel <= 8 * e_torch + 1e-6, f"fused FFN error {e_kernel:.3e} vs torch-fp32 error {e_torch:.3e} (both vs fp64)"
    # in place (out aliases x), as the pipeline calls it
    ops.ffn_fused(xd, pk, dev(b2), Hd, ops.ACT_RELU if act == "relu" else ops.ACT_GELU,
                  ln_in=(dev(gam), dev(bet)) if ln in ("in", "both") else None,
                  ln_out=(dev(gam2), dev(bet2)) if ln in ("out", "both") else None)
    assert torch.equal(xd, out)


@pytest.mark.parametrize("M,ln,plan", [(18000, "out", 2), (17999, "none", 2), (40800, "both", 2), (36800, "out", 2), (16000, "out", 1),
                                       (4600, "out", 1), (24100, "out", 0), (72000, "out", 0), (192800, "out", 0), (85570, "out", 0)])
def test_ffn_fused_split(ops, M, ln, plan):
    """tce_ffn_fused_split_f32 (round 5): a row block's hidden extent cut once, the pieces on different workgroups, the block's
    second arriver adds the partial sums and runs the epilogue.  Against the un-split launch (one more fp32 addition per element)
    and torch fp64 on a slice of rows; deterministic (a + b whichever piece arrives last: two runs bit-identical), the counters
    back at zero, shapes for which no split is planned (full rounds already / few rows) say so and the entry point refuses them;
    in place, ragged last block, LayerNorm on either side."""
    from tce_rvos_amd._lib import TceError, lib
    C, Hd = 256, 2048
    nws, ncnt = ops.ffn_split_need(M, C, Hd, ops.ACT_RELU)
    g = torch.Generator().manual_seed(M)
    w1 = torch.randn(Hd, C, generator=g) / math.sqrt(C)
    b1 = torch.randn(Hd, generator=g) * 0.2
    w2 = torch.randn(C, Hd, generator=g) / math.sqrt(Hd)
    b2 = torch.randn(C, generator=g) * 0.2
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    gam2, bet2 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    pk = ops.ffn_pack(dev(w1), dev(b1), dev(w2))
    kw = dict(ln_in=(dev(gam), dev(bet)) if ln == "both" else None, ln_out=(dev(gam2), dev(bet2)) if ln != "none" else None)
    if plan == 0:
        assert (nws, ncnt) == (0, 0)
        x = torch.zeros(M, C, device="cuda")
        with pytest.raises(TceError):
            ops.ffn_fused(x, pk, dev(b2), Hd, ops.ACT_RELU, split=(torch.zeros(1 << 20, device="cuda"), torch.zeros(4096, dtype=torch.int32, device="cuda")), **kw)
        return
    assert ncnt == (M + 127) // 128 and nws == ncnt * 2 * 128 * C
    x = torch.randn(M, C, generator=g)
    xd = dev(x)
    base = torch.empty_like(xd)
    ops.ffn_fused(xd, pk, dev(b2), Hd, ops.ACT_RELU, out=base, **kw)
    ws, cnt = torch.full((nws,), float("nan"), device="cuda"), torch.zeros(ncnt, dtype=torch.int32, device="cuda")
    out = torch.empty_like(xd)
    ops.ffn_fused(xd, pk, dev(b2), Hd, ops.ACT_RELU, out=out, split=(ws, cnt), **kw)
    torch.cuda.synchronize()
    assert int(cnt.abs().max()) == 0
    assert not bool(torch.isnan(out).any())
    scale = float(base.abs().max())
    assert float((out - base).abs().max()) <= 4e-6 * scale
    out2 = xd.clone()                                                        # second launch on the same counters, in place
    ops.ffn_fused(out2, pk, dev(b2), Hd, ops.ACT_RELU, split=(ws, cnt), **kw)
    assert torch.equal(out2, out) and int(cnt.abs().max()) == 0
    # other rows through the SAME workspace: a partner's partial sums read from a stale cache line would be the previous launch's
    x3 = dev(torch.flip(x, (0,)) * 0.7 + 0.1)
    base3, out3 = torch.empty_like(x3), torch.empty_like(x3)
    ops.ffn_fused(x3, pk, dev(b2), Hd, ops.ACT_RELU, out=base3, **kw)
    for _ in range(3):
        ops.ffn_fused(x3, pk, dev(b2), Hd, ops.ACT_RELU, out=out3, split=(ws, cnt), **kw)
        ops.ffn_fused(xd, pk, dev(b2), Hd, ops.ACT_RELU, out=out2, split=(ws, cnt), **kw)
    assert float((out3 - base3).abs().max()) <= 4e-6 * float(base3.abs().max()) and torch.equal(out2, out)
    rows = torch.cat([torch.arange(0, 300), torch.arange(M // 2, M // 2 + 200), torch.arange(M - 200, M)])
    xr = x[rows].double()
    y = F.layer_norm(xr, (C,), gam.double(), bet.double(), 1e-5) if ln == "both" else xr
    o = xr + F.linear(torch.relu(F.linear(y, w1.double(), b1.double())), w2.double(), b2.double())
    r64 = F.layer_norm(o, (C,), gam2.double(), bet2.double(), 1e-5) if ln != "none" else o
    e_split = (out[rows.cuda()].cpu().double() - r64).abs().max().item()
    e_base = (base[rows.cuda()].cpu().double() - r64).abs().max().item()
    assert e_split <= 2 * e_base + 1e-6, (e_split, e_base)
    with pytest.raises(TceError):                                            # too small a workspace is refused
        ops.ffn_fused(xd, pk, dev(b2), Hd, ops.ACT_RELU, out=out, split=(ws[:nws - 4], cnt), **kw)
    ops.check_range()


@pytest.mark.parametrize("M,C,Hd", [(72000, 96, 384), (3000, 96, 384), (20001, 128, 512)])
def test_ffn_fused_half_workgroups_are_bit_identical(ops, M, C, Hd):
    """The C <= 128 fused MLP as 128-row workgroups, two per CU, on the same packed stream (round 5; chosen by the launch's round
    count: 72000 rows are 1.1 rounds of 256-row workgroups that cost two): same arithmetic per row -- bit-identical to the 256-row
    form, LayerNorm prologue, GELU, ragged last block, in place."""
    from tce_rvos_amd._lib import lib
    g = torch.Generator().manual_seed(M + C)
    x = dev(torch.randn(M, C, generator=g))
    w1, b1 = torch.randn(Hd, C, generator=g) / math.sqrt(C), torch.randn(Hd, generator=g) * 0.2
    w2, b2 = torch.randn(C, Hd, generator=g) / math.sqrt(Hd), torch.randn(C, generator=g) * 0.2
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    pk = ops.ffn_pack(dev(w1), dev(b1), dev(w2))
    outs = {}
    for mode in (-1, 1):
        lib().tce_debug_ffn_set_half(mode)
        try:
            o = x.clone()
            ops.ffn_fused(o, pk, dev(b2), Hd, ops.ACT_GELU, ln_in=(dev(gam), dev(bet)))
            outs[mode] = o
        finally:
            lib().tce_debug_ffn_set_half(0)
    assert torch.equal(outs[-1], outs[1])
    xr = x[:256].double().cpu()
    ref = xr + F.linear(F.gelu(F.linear(F.layer_norm(xr, (C,), gam.double(), bet.double(), 1e-5), w1.double(), b1.double())), w2.double(), b2.double())
    close(outs[1][:256], ref.float(), 2e-4, 2e-4)
    ops.check_range()


def test_round_count_forms_at_boundary_sizes(ops):
    """The launch forms chosen by round counts (round 5) at sizes around their decision boundaries: the 3x3 convolution's mixed
    launches (complete rounds of 256-pixel workgroups + a remainder of 128-pixel ones) against the 128-pixel form everywhere, and
    the C = 96 fused MLP's own choice against the 256-row form -- bit-identical in both cases, ragged last blocks included."""
    from tce_rvos_amd._lib import lib
    g = torch.Generator().manual_seed(99)
    w_cl = dev(torch.randn(256, 2304, generator=g) / 48.0)
    pk = ops.conv3x3_pack(w_cl, 256)
    for (T, H, W) in ((1, 256, 256), (1, 257, 256), (1, 255, 257), (2, 256, 257), (1, 300, 301)):   # 65536 px = one wide round exactly, +-
        x = dev(torch.randn(T * H * W, 256, generator=g))
        auto = ops.conv3x3(x, pk, T, H, W, 256, 256)
        lib().tce_debug_conv3x3_set_waves(4)
        try:
            narrow = ops.conv3x3(x, pk, T, H, W, 256, 256)
        finally:
            lib().tce_debug_conv3x3_set_waves(0)
        assert torch.equal(auto, narrow), (T, H, W)
    w1, b1 = torch.randn(384, 96, generator=g) / 10, torch.randn(384, generator=g) * 0.2
    w2, b2 = torch.randn(96, 384, generator=g) / 20, torch.randn(96, generator=g) * 0.2
    pkf = ops.ffn_pack(dev(w1), dev(b1), dev(w2))
    for M in (65536, 65537, 65535 + 256, 131072 + 5, 32768 - 3, 98304 + 130):
        x = dev(torch.randn(M, 96, generator=g))
        auto = x.clone()
        ops.ffn_fused(auto, pkf, dev(b2), 384, ops.ACT_GELU)
        lib().tce_debug_ffn_set_half(-1)
        try:
            wide = x.clone()
            ops.ffn_fused(wide, pkf, dev(b2), 384, ops.ACT_GELU)
        finally:
            lib().tce_debug_ffn_set_half(0)
        assert torch.equal(auto, wide), M
    ops.check_range()


def test_ffn_fused_rejects_bad_arguments(ops):
    from tce_rvos_amd._lib import TceError
    with pytest.raises(ValueError):
        ops.ffn_pack(dev(torch.zeros(64, 100)), None, dev(torch.zeros(100, 64)))   # C = 100 unsupported
    w1, w2 = dev(torch.zeros(64, 96)), dev(torch.zeros(96, 64))
    pk = ops.ffn_pack(w1, None, w2)
    x = dev(torch.zeros(10, 96))
    with pytest.raises(TceError):
        ops.ffn_fused(x, pk, dev(torch.zeros(96)), 64, 0)   # no activation code 0


def test_split_fp16_range_guard(ops):
    """The split-fp16 GEMM's operand contract (|x| inside the fp16 range) is guarded: weights on the host when they are
    packed, activations by the sticky device flag every split-mode epilogue raises when it stores |v| >= 60000 / Inf /
    NaN.  Tiny operands are not an error: the documented absolute floor (6e-8 per factor) must hold."""
    from tce_rvos_amd.ops import RangeError
    g = torch.Generator().manual_seed(5)
    M, N, K = 300, 256, 256
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    ops.check_range()                                   # clean start
    ops.gemm(dev(a), dev(w))
    ops.check_range()                                   # in-range problem: no flag
    out = ops.gemm(dev(a * 1e5), dev(w))                # result ~1e5: outside the range the NEXT GEMM could consume
    with pytest.raises(RangeError):
        ops.check_range()
    ops.check_range()                                   # the flag is cleared by the raise
    nan = a.clone()
    nan[7, 3] = float("nan")
    ops.gemm(dev(nan), dev(w))
    with pytest.raises(RangeError):
        ops.check_range()
    with pytest.raises(RangeError):                     # host half: weights beyond the range are refused at pack time
        ops.check_weight_range([("w", dev(w * 1e6))])
    ops.check_weight_range([("w", dev(w))])
    # tiny operands: absolute error within K * 6e-8 * max|w| * (a few) of the fp64 result -- never silently large
    tiny = a * 1e-6
    out = ops.gemm(dev(tiny), dev(w)).cpu().double()
    ref = tiny.double() @ w.double().T
    assert (out - ref).abs().max().item() <= 4 * 6e-8 * w.abs().max().item() * math.sqrt(K)
    ops.check_range()
    # the fused FFN raises the flag for an out-of-range hidden value as well
    C, Hd = 256, 64
    w1, w2 = torch.randn(Hd, C, generator=g), torch.randn(C, Hd, generator=g) / 8
    pk = ops.ffn_pack(dev(w1), dev(torch.zeros(Hd)), dev(w2))
    x = dev(torch.randn(200, C, generator=g) * 1e4)
    ops.ffn_fused(x, pk, dev(torch.zeros(C)), Hd, ops.ACT_RELU, out=torch.empty_like(x))
    with pytest.raises(RangeError):
        ops.check_range()


@pytest.mark.parametrize("M,N,K,batch", [(24100, 256, 256, 1), (4820, 384, 256, 5), (7200, 288, 96, 1), (1000, 576, 192, 2),
                                         (333, 128, 128, 1), (129, 32, 96, 1),
                                         (4600, 1152, 384, 1), (4600, 384, 384, 1), (920, 1536, 384, 5), (200, 256, 384, 1),
                                         (16200, 512, 512, 1), (700, 1536, 512, 3), (130, 256, 512, 1)])
@pytest.mark.parametrize("variant", ["plain", "a2_relu", "res_mul", "gelu_res", "ln_in", "ln_out"])
def test_rowlin(ops, M, N, K, batch, variant):
    """Token-stationary linear kernel (tce_rowlin_f32) against torch fp32 of the op sequence it replaces: addend with a
    shared (stride-0) position map, LayerNorm prologue, activations, residual add / multiply, LayerNorm epilogue,
    frame-batched launches with per-frame strides, ragged row counts."""
    if variant == "ln_out" and N != 256:
        pytest.skip("output LayerNorm is built for N = 256")
    if variant == "ln_out" and K == 512:   # x alone fills half the register file at K = 512: the entry point must refuse
        from tce_rvos_amd._lib import TceError
        with pytest.raises(TceError):
            ops.rowlin(torch.empty(M, K, device="cuda"), ops.rowlin_pack(torch.zeros(N, K, device="cuda")),
                       torch.empty(M, N, device="cuda"), M, N, K, K, N, ln_out=(torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")),
                       res=torch.zeros(M, N, device="cuda"), ldres=N, res_mode=ops.RES_ADD)
        return
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(batch, M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.3
    pos = torch.randn(M, K, generator=g)
    res = torch.randn(batch, M, N, generator=g)
    gi, bi = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    go, bo = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.2
    pk = ops.rowlin_pack(dev(w))
    dx, dres = dev(x), dev(res)
    out = torch.empty(batch, M, N, device="cuda")
    kw = dict(bias=dev(b), batch=batch, sX=M * K, sOut=M * N)
    if variant == "plain":
        ref = F.linear(x, w, b)
    elif variant == "a2_relu":
        kw.update(a2=dev(pos), lda2=K, sA2=0, act=ops.ACT_RELU)
        ref = F.relu(F.linear(x + pos, w, b))
    elif variant == "res_mul":
        kw.update(res=dres, ldres=N, res_mode=ops.RES_MUL, sRes=M * N)
        ref = F.linear(x, w, b) * res
    elif variant == "gelu_res":
        kw.update(act=ops.ACT_GELU, res=dres, ldres=N, res_mode=ops.RES_ADD, sRes=M * N)
        ref = F.gelu(F.linear(x, w, b)) + res
    elif variant == "ln_in":
        kw.update(ln_in=(dev(gi), dev(bi)))
        ref = F.linear(F.layer_norm(x, (K,), gi, bi, 1e-5), w, b)
    else:
        kw.update(res=dres, ldres=N, res_mode=ops.RES_ADD, sRes=M * N, ln_out=(dev(go), dev(bo)))
        ref = F.layer_norm(F.linear(x, w, b) + res, (N,), go, bo, 1e-5)
    ops.rowlin(dx, pk, out, M, N, K, K, N, **kw)
    close(out, ref, 2e-4, 2e-4)
    if variant == "plain" and batch == 1:   # position map shared by groups of rows inside ONE launch (a2_rows)
        rows = 97
        ops.rowlin(dx, pk, out, M, N, K, K, N, bias=dev(b), a2=dev(pos[:rows]), lda2=K, a2_rows=rows)
        close(out[0], F.linear(x[0] + pos[:rows][torch.arange(M) % rows], w, b), 2e-4, 2e-4)


def test_msda_fused_lds_staged_is_bit_identical(ops):
    """Encoder-sized call (thousands of queries per frame): the LDS-staged kernel (coarse levels of a (frame, head) slice
    in LDS) against the L2-gather kernel it replaces -- same arithmetic, same order: bit-identical -- and against the
    oracle's restatement of the reference core."""
    from tce_rvos_amd._lib import lib
    g = torch.Generator().manual_seed(77)
    N, M, L, P = 2, 8, 4, 4
    shapes = [(45, 80), (23, 40), (12, 20), (6, 10)]   # config 2's levels
    S = sum(h * w for h, w in shapes)
    Lq = S
    value = torch.randn(N, S, M, 32, generator=g)
    proj = torch.randn(N, Lq, M * L * P * 3, generator=g)
    proj[..., :M * L * P * 2] *= 3.0
    ref = torch.rand(Lq, 2, generator=g) * 1.1 - 0.05
    dv, dp, dr = dev(value), dev(proj), dev(ref)
    b = ops.msda_fused(dv, dp, dr, shapes, N, S, M, Lq, L, P, 2, False)
    lib().tce_debug_msda_set_lds(1)
    try:
        a = ops.msda_fused(dv, dp, dr, shapes, N, S, M, Lq, L, P, 2, False)
    finally:
        lib().tce_debug_msda_set_lds(0)
    assert torch.equal(a, b)
    # round 5: the default form keeps four sampling points in flight (branch-free corner addresses, zero coefficients for absent
    # corners); the one-point-at-a-time loop (2) and two in flight (3) give the same bits
    for mode in (2, 3):
        lib().tce_debug_msda_set_lds(mode)
        try:
            c = ops.msda_fused(dv, dp, dr, shapes, N, S, M, Lq, L, P, 2, False)
        finally:
            lib().tce_debug_msda_set_lds(0)
        assert torch.equal(c, b), mode
    off = proj[..., :M * L * P * 2].view(N, Lq, M, L, P, 2)
    aw = torch.softmax(proj[..., M * L * P * 2:].view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    norm = torch.tensor([[w, h] for (h, w) in shapes], dtype=torch.float32)
    loc = ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
    close(a.view(N, Lq, M * 32), O.msda_core(value, shapes, loc, aw), 1e-4, 1e-4)


def test_single_pass_fp16_mode(ops):
    """tce_set_gemm_mode(2) (BASELINE config 5's "fp16 MFMA"): one MFMA per product on operands rounded to nearest fp16,
    fp32 accumulation -- in the tiled GEMM, the fused FFN and the token-stationary linear kernel alike.  Its error sits
    where fp16 rounding of the operands puts it (~1e-3 of sum|a||b|), three orders above the default 3 x fp16 split."""
    g = torch.Generator().manual_seed(3)
    M, N, K = 2000, 256, 256
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    ref = a.double() @ w.double().T + b.double()
    scale = (a.double().abs() @ w.double().abs().T)
    errs = {}
    try:
        for mode in ("f16x3", "f16"):
            ops.set_gemm_mode(mode)
            o1 = ops.gemm(dev(a), dev(w), bias=dev(b)).cpu().double()
            pk = ops.rowlin_pack(dev(w))
            o2 = torch.empty(M, N, device="cuda")
            ops.rowlin(dev(a), pk, o2, M, N, K, K, N, bias=dev(b))
            errs[mode] = (((o1 - ref).abs() / scale).max().item(), ((o2.cpu().double() - ref).abs() / scale).max().item())
            w1, w2 = torch.randn(512, K, generator=g) / 16, torch.randn(K, 512, generator=g) / 22
            fpk = ops.ffn_pack(dev(w1), dev(torch.zeros(512)), dev(w2))
            of = torch.empty(M, K, device="cuda")
            ops.ffn_fused(dev(a), fpk, dev(torch.zeros(K)), 512, ops.ACT_RELU, out=of)
            rf = a.double() + torch.relu(a.double() @ w1.double().T) @ w2.double().T
            errs[mode] += ((of.cpu().double() - rf).abs().max().item() / rf.abs().max().item(),)
            # pixel-stationary 3x3 convolution and MFMA patch embedding follow the mode as well
            xc, wc = torch.randn(2, 256, 12, 20, generator=g), torch.randn(256, 256, 3, 3, generator=g) / 48
            rc = F.conv2d(xc.double(), wc.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, 256)
            wcl = dev(wc.permute(0, 2, 3, 1).reshape(256, -1).contiguous())
            oc = ops.conv3x3(dev(xc.permute(0, 2, 3, 1).reshape(-1, 256).contiguous()), ops.conv3x3_pack(wcl, 256), 2, 12, 20,
                             256, 256)
            errs[mode] += ((oc.cpu().double() - rc).abs().max().item() / rc.abs().max().item(),)
            xp, wp = torch.randn(1, 3, 32, 48, generator=g), torch.randn(96, 3, 4, 4, generator=g) / 7
            rp = F.layer_norm(F.conv2d(xp.double(), wp.double(), stride=4).flatten(2).transpose(1, 2), (96,)).reshape(-1, 96)
            op_, _, _ = ops.patch_embed(dev(xp), dev(wp), dev(torch.zeros(96)), dev(torch.ones(96)), dev(torch.zeros(96)))
            errs[mode] += ((op_.cpu().double() - rp).abs().max().item() / rp.abs().max().item(),)
    finally:
        ops.set_gemm_mode("f16x3")
    print("errors (gemm, rowlin: relative to sum|a||b|; ffn, conv3x3, patch embed: relative to max|out|):", errs)
    for k in range(5):
        assert errs["f16x3"][k] < 3e-6
        assert 1e-5 < errs["f16"][k] < 2e-3


@pytest.mark.parametrize("M,L,mode", [(18000, 32, "add_ln"), (4600, 11, "add_ln"), (301, 32, "mul"), (7200, 20, "mul_batched")])
def test_xattn_fused(ops, M, L, mode):
    """Text cross-attention (q-proj -> 8-head attention over L <= 32 keys -> out-proj -> residual -> LayerNorm) as one
    token-stationary launch against nn.MultiheadAttention semantics in torch fp32 (segmentation.py:366-371 with a
    position addend on the query, :455-464 with the multiplicative residual)."""
    g = torch.Generator().manual_seed(M + L)
    Cn = 256
    x = torch.randn(M, Cn, generator=g)
    pos = torch.randn(M // 5 if M % 5 == 0 else M, Cn, generator=g) * 0.5
    text = torch.randn(L, Cn, generator=g)
    mha = torch.nn.MultiheadAttention(Cn, 8)
    with torch.no_grad():
        for p_ in mha.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.06 if p_.dim() == 2 else 0.2))
    Wq, Wk, Wv = mha.in_proj_weight.detach().chunk(3, 0)
    bq, bk, bv = mha.in_proj_bias.detach().chunk(3, 0)
    k = F.linear(text, Wk, bk)
    v = F.linear(text, Wv, bv)
    gam, bet = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.2
    rows = pos.shape[0]
    qin = x + pos[torch.arange(M) % rows] if mode == "add_ln" else x
    with torch.no_grad():
        att = mha(qin[:, None], text[:, None], text[:, None])[0][:, 0]
    if mode == "add_ln":
        ref = F.layer_norm(x + att, (Cn,), gam, bet, 1e-5)
    else:
        ref = x * att
    ar = lambda *shape, dtype=torch.float32: torch.empty(*shape, dtype=dtype, device="cuda")
    wqT = ops.xattn_static(dev(Wq), dev(bq))
    pk = ops.xattn_pack(dev(k), dev(v), wqT, dev(mha.out_proj.weight.detach()), L, ar)
    xd = dev(x)
    out = torch.empty_like(xd)
    bo = dev(mha.out_proj.bias.detach())
    if mode == "add_ln":
        ops.xattn_fused(xd, pk, bo, M, out, a2=dev(pos), a2_rows=rows if rows != M else 0, ln_out=(dev(gam), dev(bet)))
    elif mode == "mul":
        ops.xattn_fused(xd, pk, bo, M, out, res_mode=ops.RES_MUL)
    else:  # frame-batched: 4 frames of M/4 rows written into a wider destination with a per-frame stride
        nb, mb = 4, M // 4
        wide = torch.zeros(nb, mb + 50, Cn, device="cuda")
        ops.xattn_fused(xd, pk, bo, mb, wide, res_mode=ops.RES_MUL, batch=nb, sX=mb * Cn, sRes=mb * Cn, sOut=(mb + 50) * Cn)
        out = wide[:, :mb].reshape(M, Cn)
        assert float(wide[:, mb:].abs().max()) == 0.0
    close(out, ref, 3e-4, 3e-4)


@pytest.mark.parametrize("M,L,group,batch", [(18000, 32, 32, 1), (4820, 8, 8, 5), (301, 11, 32, 1), (130, 8, 8, 3)])
def test_xattn_ffn_chain_matches_two_launches(ops, M, L, group, batch):
    """tce_xattn_ffn_fused_f32 (round 5): cross-attention -> LayerNorm -> FFN -> LayerNorm as ONE launch against the two launches
    it replaces (tce_xattn_fused_f32, then tce_ffn_fused_f32 in place): the attention stage's rows land in `mid` bit-identical,
    the FFN stage (operand taken from the accumulator registers, W1 packed in their k order) within fp32 round-off of the
    two-launch result and of torch fp64; per-frame weight streams (group 8, batch = frames) and the text form (group 32); ragged
    last row block; in place on x."""
    g = torch.Generator().manual_seed(M + L)
    Cn, Hd = 256, 2048
    x = torch.randn(batch * M, Cn, generator=g)
    pos = torch.randn(M, Cn, generator=g) * 0.5
    k, v = torch.randn(batch, L, Cn, generator=g), torch.randn(batch, L, Cn, generator=g)
    Wq, bq = torch.randn(Cn, Cn, generator=g) * 0.06, torch.randn(Cn, generator=g) * 0.2
    Wo, bo = torch.randn(Cn, Cn, generator=g) * 0.06, torch.randn(Cn, generator=g) * 0.2
    g1, be1 = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.2
    W1, b1 = torch.randn(Hd, Cn, generator=g) / 16, torch.randn(Hd, generator=g) * 0.2
    W2, b2 = torch.randn(Cn, Hd, generator=g) / 45, torch.randn(Cn, generator=g) * 0.2
    g2, be2 = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.2
    ar = lambda *shape, dtype=torch.float32: torch.empty(*shape, dtype=dtype, device="cuda")
    wqT = ops.xattn_static(dev(Wq), dev(bq))
    pk = ops.xattn_pack(dev(k), dev(v), wqT, dev(Wo), L, ar, group=group, batch=batch)
    kw = dict(a2=dev(pos), ln_out=(dev(g1), dev(be1)), batch=batch, sX=M * Cn, sOut=M * Cn, group=group, per_batch_weights=batch > 1)
    # the two launches
    y = torch.empty(batch * M, Cn, device="cuda")
    ops.xattn_fused(dev(x), pk, dev(bo), M, y, **kw)
    ref2 = y.clone()
    ops.ffn_fused(ref2, ops.ffn_pack(dev(W1), dev(b1), dev(W2)), dev(b2), Hd, ops.ACT_RELU, ln_out=(dev(g2), dev(be2)))
    # the chain, in place on x
    xd, mid = dev(x), torch.full((batch * M, Cn), float("nan"), device="cuda")
    ops.xattn_fused(xd, pk, dev(bo), M, xd, ffn=(ops.ffn_pack_chain(dev(W1), dev(b1), dev(W2)), dev(b2), Hd, (dev(g2), dev(be2)), mid, M * Cn), **kw)
    torch.cuda.synchronize()
    assert torch.equal(mid, y)                                   # stage 1 is the same arithmetic
    scale = float(ref2.abs().max())
    assert float((xd - ref2).abs().max()) <= 2e-5 * scale        # stage 2: another summation order of the first product only
    yd = y.double().cpu()
    ref64 = F.layer_norm(yd + F.linear(F.relu(F.linear(yd, W1.double(), b1.double())), W2.double(), b2.double()), (Cn,), g2.double(), be2.double(), 1e-5)
    assert float((xd.double().cpu() - ref64).abs().max()) <= 3e-5 * scale
    ops.check_range()


@pytest.mark.parametrize("L,group,batch", [(32, 32, 1), (9, 32, 1), (8, 8, 5), (5, 8, 2)])
def test_xattn_pack_one_launch_is_bit_identical(ops, L, group, batch):
    """tce_xattn_pack_f32 (fold + pack in one launch) against tce_xattn_prepare_f32 -> tce_ffn_pack_batched_f32: the same bytes."""
    g = torch.Generator().manual_seed(L + group + batch)
    k, v = dev(torch.randn(batch * L, 256, generator=g)), dev(torch.randn(batch * L, 256, generator=g))
    wq, bq = dev(torch.randn(256, 256, generator=g) * 0.06), dev(torch.randn(256, generator=g) * 0.2)
    wo = dev(torch.randn(256, 256, generator=g) * 0.06)
    wqT = ops.xattn_static(wq, bq)
    ar = lambda *shape, dtype=torch.float32: torch.zeros(*shape, dtype=dtype, device="cuda")
    saved = ops.XATTN_PACK_FUSED
    try:
        ops.XATTN_PACK_FUSED = True
        a = ops.xattn_pack(k, v, wqT, wo, L, ar, group=group, batch=batch)
        ops.XATTN_PACK_FUSED = False
        b = ops.xattn_pack(k, v, wqT, wo, L, ar, group=group, batch=batch)
    finally:
        ops.XATTN_PACK_FUSED = saved
    torch.cuda.synchronize()
    assert a.shape == b.shape and torch.equal(a, b)


def test_xattn_fused_eight_key_groups_per_frame(ops):
    """FrameTokenLayer's pixel <- token attention (tce_deformable_transformer.py:480-484): every frame has its own 8 keys /
    values, so the folded weights differ per batch entry; softmax over groups of 8 inside a 32-unit chunk."""
    g = torch.Generator().manual_seed(99)
    T, S, F_, Cn = 3, 1500, 8, 256
    x = torch.randn(T, S, Cn, generator=g)
    pos = torch.randn(S, Cn, generator=g) * 0.5
    tok = torch.randn(T, F_, Cn, generator=g)
    mha = torch.nn.MultiheadAttention(Cn, 8)
    with torch.no_grad():
        for p_ in mha.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.06 if p_.dim() == 2 else 0.2))
    Wq, Wk, Wv = mha.in_proj_weight.detach().chunk(3, 0)
    bq, bk, bv = mha.in_proj_bias.detach().chunk(3, 0)
    gam, bet = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.2
    with torch.no_grad():   # seq-first: (L = S, N = T) queries against (F, T) keys
        att = mha((x + pos).permute(1, 0, 2), tok.permute(1, 0, 2), tok.permute(1, 0, 2))[0].permute(1, 0, 2)
    ref = F.layer_norm(x + att, (Cn,), gam, bet, 1e-5)
    k = F.linear(tok, Wk, bk).reshape(T * F_, Cn)
    v = F.linear(tok, Wv, bv).reshape(T * F_, Cn)
    ar = lambda *shape, dtype=torch.float32: torch.empty(*shape, dtype=dtype, device="cuda")
    pk = ops.xattn_pack(dev(k), dev(v), ops.xattn_static(dev(Wq), dev(bq)), dev(mha.out_proj.weight.detach()), F_, ar,
                        group=8, batch=T)
    xd = dev(x.reshape(T * S, Cn))
    ops.xattn_fused(xd, pk, dev(mha.out_proj.bias.detach()), S, xd, a2=dev(pos), ln_out=(dev(gam), dev(bet)), batch=T,
                    sX=S * Cn, sOut=S * Cn, group=8, per_batch_weights=True)
    close(xd.view(T, S, Cn), ref, 3e-4, 3e-4)


def test_copy_segments_plan_and_many(ops):
    """tce_copy_segments: dense, misaligned, strided-row and int64 sources in one launch; > 16 segments split."""
    g = torch.Generator(device="cpu").manual_seed(5)
    base = torch.randn(40000, generator=g).cuda()
    srcs = [base[:1000], base[1001:1008], base[2000:2600].view(1, 5, 30, 4)[..., :2], base[3:4],
            torch.arange(37, device="cuda"), base[8192:8192 + 20000].view(100, 200)]
    srcs += [base[i * 13:i * 13 + 11] for i in range(20)]
    plan = ops.CopyPlan(srcs)
    outs = plan.clone()
    torch.cuda.synchronize()
    for o, s_ in zip(outs, srcs):
        assert o.shape == s_.shape and o.dtype == s_.dtype and o.is_contiguous()
        assert torch.equal(o, s_)
    dsts = [torch.zeros_like(s_, memory_format=torch.contiguous_format) for s_ in srcs]
    ops.copy_many(dsts, srcs)
    torch.cuda.synchronize()
    for o, s_ in zip(dsts, srcs):
        assert torch.equal(o, s_)
    with pytest.raises(ValueError):
        ops.CopyPlan([base[:100:3]])  # neither dense nor a row gather


@pytest.fixture(params=[4, 8, 0])
def conv_waves(request):
    """The forms of the pixel-stationary convolution: 128-pixel (4-wave) workgroups, 256-pixel (8-wave, round 5) ones, and the
    launcher's own choice (0) -- at 72000 pixels the mixed form: one round of 256-pixel workgroups, the remainder as 128-pixel ones."""
    from tce_rvos_amd._lib import lib
    lib().tce_debug_conv3x3_set_waves(request.param)
    yield request.param
    lib().tce_debug_conv3x3_set_waves(0)


@pytest.mark.parametrize("T,H,W", [(1, 9, 13), (2, 32, 40), (1, 45, 80), (3, 17, 5)])
def test_conv3x3_pixel_stationary(ops, conv_waves, T, H, W):
    """tce_conv3x3_f32 (pixel-stationary kernel) against torch conv2d in fp64 and against the implicit-GEMM path;
    image borders, a ragged last pixel block and pixels whose 3x3 neighbourhood crosses frames."""
    g = torch.Generator(device="cpu").manual_seed(11 + H)
    x = torch.randn(T, 256, H, W, generator=g)
    w = torch.randn(256, 256, 3, 3, generator=g) / 48.0
    b = torch.randn(256, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(T * H * W, 256)
    x_cl = x.permute(0, 2, 3, 1).reshape(T * H * W, 256).contiguous().cuda()
    w_cl = w.permute(0, 2, 3, 1).reshape(256, -1).contiguous().cuda()
    pk = ops.conv3x3_pack(w_cl, 256)
    out = ops.conv3x3(x_cl, pk, T, H, W, 256, 256, bias=b.cuda())
    gem, _, _ = ops.conv2d_cl(x_cl, w_cl, T, H, W, 256, 3, 3, 1, 1, bias=b.cuda())
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    assert (out.double().cpu() - ref).abs().max().item() < 2e-6 * scale * 8
    assert (out - gem).abs().max().item() < 2e-6 * scale * 8


# ---------------------------------------------------------------------------------------------------------------
# Cold-operand variants (VERDICT r2 weak #9 / next #8a).  The conv3x3 kernel once mixed LDS-DMA and register loads under
# one counted vmcnt: right on cache-resident operands (every stand-alone test), wrong inside the clip where operands
# come from HBM with uneven latency.  Here the operands are larger than the L2s (8 x 4 MiB), and the L2s + the 256 MiB
# memory-side cache are flushed by a 768 MiB write between producing the operands and the launch; EVERY output row is
# checked (the failure was in lanes 1-7 of every 8), against fp64 on the GPU, three launches each (the failure was
# intermittent).
# ---------------------------------------------------------------------------------------------------------------
def _evict_caches():
    junk = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")
    junk.fill_(1)
    junk.fill_(2)
    torch.cuda.synchronize()
    del junk


def test_conv3x3_cold_operands(ops, conv_waves):
    T, H, W = 5, 90, 160   # config 2's stride-4 map: 72000 pixels x 1 KiB = 73.7 MB in, 73.7 MB out
    g = torch.Generator(device="cpu").manual_seed(5)
    x_cl = torch.randn(T * H * W, 256, generator=g).cuda()
    w = torch.randn(256, 256, 3, 3, generator=g) / 48.0
    b = torch.randn(256, generator=g).cuda()
    w_cl = w.permute(0, 2, 3, 1).reshape(256, -1).contiguous().cuda()
    pk = ops.conv3x3_pack(w_cl, 256)
    x64 = x_cl.view(T, H, W, 256).permute(0, 3, 1, 2).double()
    ref = torch.nn.functional.conv2d(x64, w.cuda().double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(T * H * W, 256)
    scale = ref.abs().max().item()
    out = torch.empty(T * H * W, 256, device="cuda")
    for rep in range(3):
        out.fill_(float("nan"))
        _evict_caches()
        ops.conv3x3(x_cl, pk, T, H, W, 256, 256, bias=b, out=out)
        torch.cuda.synchronize()
        err = (out.double() - ref).abs().amax(dim=1)
        bad = int((~(err < 2e-5 * scale)).sum())
        assert bad == 0, f"launch {rep}: {bad} of {T * H * W} pixels wrong (max err {err.nan_to_num(1e30).max().item():.3e})"


@pytest.mark.parametrize("M,C,Hd,act", [(72000, 256, 2048, "relu"), (72000, 96, 384, "gelu")])
def test_ffn_fused_cold_operands(ops, M, C, Hd, act):
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g).cuda()
    w1 = (torch.randn(Hd, C, generator=g) / math.sqrt(C)).cuda()
    b1 = (torch.randn(Hd, generator=g) * 0.2).cuda()
    w2 = (torch.randn(C, Hd, generator=g) / math.sqrt(Hd)).cuda()
    b2 = (torch.randn(C, generator=g) * 0.2).cuda()
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    pk = ops.ffn_pack(w1, b1, w2)
    ln_in = act == "gelu"
    with torch.no_grad():
        ref = torch.empty(M, C, dtype=torch.float64, device="cuda")
        for lo in range(0, M, 8000):   # fp64 reference in row blocks (the hidden tensor is M x Hd doubles)
            xx = x[lo:lo + 8000].double()
            y = F.layer_norm(xx, (C,), gam.double(), bet.double(), 1e-5) if ln_in else xx
            hdn = F.linear(y, w1.double(), b1.double())
            hdn = F.gelu(hdn) if act == "gelu" else torch.relu(hdn)
            o = xx + F.linear(hdn, w2.double(), b2.double())
            ref[lo:lo + 8000] = o if ln_in else F.layer_norm(o, (C,), gam.double(), bet.double(), 1e-5)
    out = torch.empty_like(x)
    for rep in range(3):
        out.fill_(float("nan"))
        _evict_caches()
        ops.ffn_fused(x, pk, b2, Hd, ops.ACT_GELU if act == "gelu" else ops.ACT_RELU,
                      ln_in=(gam, bet) if ln_in else None, ln_out=None if ln_in else (gam, bet), out=out)
        torch.cuda.synchronize()
        err = (out.double() - ref).abs().amax(dim=1)
        bad = int((~(err < 1e-4 * max(1.0, ref.abs().max().item()))).sum())   # a mis-ordered load gives O(0.1 .. 1)
        assert bad == 0, f"launch {rep}: {bad} of {M} rows wrong (max err {err.nan_to_num(1e30).max().item():.3e})"


@pytest.mark.parametrize("M,N,K,variant", [(72000, 256, 256, "ln_out"), (72000, 384, 256, "plain"), (72000, 288, 96, "ln_in"),
                                           (4600, 1152, 384, "ln_in"), (7680, 384, 384, "plain"),     # round 4: K = 384 (Swin-T stage 3)
                                           (16200, 512, 512, "plain"), (16200, 1536, 512, "ln_in")])  # and K = 512 (Swin-B stage 3)
def test_rowlin_cold_operands(ops, M, N, K, variant):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
    b = (torch.randn(N, generator=g) * 0.3).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    gi, bi = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.2).cuda()
    go, bo = (torch.rand(N, generator=g) + 0.5).cuda(), (torch.randn(N, generator=g) * 0.2).cuda()
    pk = ops.rowlin_pack(w)
    kw = dict(bias=b)
    x64 = x.double()
    if variant == "plain":
        ref = F.linear(x64, w.double(), b.double())
    elif variant == "ln_in":
        kw.update(ln_in=(gi, bi))
        ref = F.linear(F.layer_norm(x64, (K,), gi.double(), bi.double(), 1e-5), w.double(), b.double())
    else:
        kw.update(res=res, ldres=N, res_mode=ops.RES_ADD, ln_out=(go, bo))
        ref = F.layer_norm(F.linear(x64, w.double(), b.double()) + res.double(), (N,), go.double(), bo.double(), 1e-5)
    out = torch.empty(M, N, device="cuda")
    for rep in range(3):
        out.fill_(float("nan"))
        _evict_caches()
        ops.rowlin(x, pk, out, M, N, K, K, N, **kw)
        torch.cuda.synchronize()
        err = (out.double() - ref).abs().amax(dim=1)
        bad = int((~(err < 1e-4 * max(1.0, ref.abs().max().item()))).sum())   # a mis-ordered load gives O(0.1 .. 1)
        assert bad == 0, f"launch {rep}: {bad} of {M} rows wrong (max err {err.nan_to_num(1e30).max().item():.3e})"


@pytest.mark.parametrize("R,K", [(40, 256), (25, 256), (1, 256), (64, 256), (200, 256), (32, 768), (7, 96)])
def test_fewrow_linear(ops, R, K):
    """tce_fewrow_linear_f32 (exact fp32): three projections of the same rows in one launch (position map on two of them,
    ReLU / sigmoid / GELU), the residual of segment 0 in place, ragged slabs (N = 2, 384, 300), several 32-row passes."""
    g = torch.Generator().manual_seed(R * 7 + K)
    x = torch.randn(R, K, generator=g)
    pos = torch.randn(8, K, generator=g)              # shared by groups of 8 rows (the frame's tokens)
    w1, b1 = torch.randn(384, K, generator=g) / math.sqrt(K), torch.randn(384, generator=g) * 0.2
    w2, b2 = torch.randn(2, K, generator=g) / math.sqrt(K), torch.randn(2, generator=g) * 0.2
    w3 = torch.randn(300, K, generator=g) / math.sqrt(K)
    dx, dpos = dev(x), dev(pos)
    o1, o2, o3 = (torch.full((R, n), float("nan"), device="cuda") for n in (384, 2, 300))
    ops.fewrow_linear(dx, R, K, [(dev(w1), dev(b1), o1, 384, 384, True, ops.FR_NONE),
                                 (dev(w2), dev(b2), o2, 2, 2, False, ops.FR_SIGMOID),
                                 (dev(w3), None, o3, 300, 300, True, ops.FR_RELU)], a2=dpos, lda2=K, a2_rows=8)
    xp = x + pos[torch.arange(R) % 8]
    close(o1, F.linear(xp, w1, b1), 2e-5, 2e-5)
    close(o2, torch.sigmoid(F.linear(x, w2, b2)), 2e-5, 2e-5)
    close(o3, F.relu(F.linear(xp, w3)), 2e-5, 2e-5)
    # out_proj + residual in place on the residual stream (x is another tensor), N = 256 and a ragged 300
    for N in (256, 300):
        wo, bo = torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g) * 0.2
        tok = torch.randn(R, N, generator=g)
        dtok = dev(tok)
        ops.fewrow_linear(dx, R, K, [(dev(wo), dev(bo), dtok, N, N, False, ops.FR_GELU if N == 300 else ops.FR_NONE)], res=dtok,
                          ldres=N)
        lin = F.linear(x, wo, bo)
        close(dtok, tok + (F.gelu(lin) if N == 300 else lin), 5e-5, 5e-5)
    from tce_rvos_amd._lib import TceError
    with pytest.raises(TceError):   # an output on top of the rows other workgroups still read
        ops.fewrow_linear(dx, R, K, [(dev(torch.randn(K, K)), None, dx, K, K, False, ops.FR_NONE)])
    if K == 256:
        # LayerNorm PROLOGUE (round 5): the rows are normalised before the addend and the projections, and written out once
        ga, be = 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
        xs = x * 3.0 + 0.7                                  # not already normalised
        xn = F.layer_norm(xs, (K,), ga, be, 1e-5)
        o1.fill_(float("nan")); o3.fill_(float("nan"))
        xn_out = torch.full((R, K), float("nan"), device="cuda")
        ops.fewrow_linear(dev(xs), R, K, [(dev(w1), dev(b1), o1, 384, 384, True, ops.FR_NONE), (dev(w3), None, o3, 300, 300, False, ops.FR_RELU)],
                          a2=dpos, lda2=K, a2_rows=8, ln_in=(dev(ga), dev(be)), xn_out=xn_out)
        close(xn_out, xn, 2e-5, 2e-5)
        close(o1, F.linear(xn + pos[torch.arange(R) % 8], w1, b1), 5e-5, 5e-5)
        close(o3, F.relu(F.linear(xn, w3)), 5e-5, 5e-5)
        with pytest.raises(TceError):   # the normalised copy on top of the rows other workgroups still read
            dxs = dev(xs)
            ops.fewrow_linear(dxs, R, K, [(dev(w1), dev(b1), o1, 384, 384, False, ops.FR_NONE)], ln_in=(dev(ga), dev(be)), xn_out=dxs)
    else:
        with pytest.raises(TceError):   # the prologue needs the whole row staged at once (K = 256)
            ops.fewrow_linear(dx, R, K, [(dev(w1), dev(b1), o1, 384, 384, False, ops.FR_NONE)], ln_in=(dev(torch.ones(K)), dev(torch.zeros(K))))


@pytest.mark.parametrize("M,N,K,splits,res", [(32, 768, 3072, 16, True), (25, 256, 2048, 8, True), (1, 768, 768, 4, False),
                                              (40, 1024, 512, 2, True), (32, 768, 768, 1, True)])
def test_gemm_splitk_with_layernorm(ops, M, N, K, splits, res):
    """tce_gemm_splitk_ln_f32: the LayerNorm of a post-norm block folded into the split-K reduction pass (in place on the
    residual stream, as the text encoder and the decoder FFN call it); splits = 1 takes GEMM + LayerNorm launches."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.2
    x = torch.randn(M, N, generator=g)
    gam, bet = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.2
    ref = F.layer_norm(F.linear(a, w, b) + (x if res else 0), (N,), gam, bet, 1e-12)
    dx = dev(x)
    ws = torch.empty(max(1, splits) * M * N, device="cuda")
    ops.gemm_ex(dev(a), dev(w), dx, M, N, K, K, K, N, bias=dev(b), res=dx if res else None, ldres=N,
                res_mode=ops.RES_ADD if res else ops.RES_NONE, splitk=splits, ws=ws if splits > 1 else None,
                ln=(dev(gam), dev(bet)), ln_eps=1e-12)
    close(dx, ref, 2e-4, 2e-4)


@pytest.mark.parametrize("N,Lq,ref_dim", [(2, 300, 2), (3, 5, 4), (2, 700, 2)])
def test_msda_fused_padded_levels(ops, N, Lq, ref_dim):
    """tce_msda_fused_valid_f32: reference points scaled per level by the valid ratios (tce_deformable_transformer.py:125-132,
    590-594, 654-656) and the value rows of padded positions read as zero (ms_deform_attn.py:96-97) -- both kernel forms."""
    g = torch.Generator().manual_seed(Lq + ref_dim)
    M, L, P = 8, 4, 4
    shapes = [(9, 13), (5, 7), (3, 4), (2, 2)]
    valid = [(7, 10), (4, 5), (2, 3), (1, 2)]
    S = sum(h * w for h, w in shapes)
    value = torch.randn(N, S, M, 32, generator=g)
    proj = torch.randn(N, Lq, M * L * P * 3, generator=g)
    proj[..., :M * L * P * 2] *= 2.0
    ref = torch.rand(N, Lq, ref_dim, generator=g) * 1.2 - 0.1
    off = proj[..., :M * L * P * 2].view(N, Lq, M, L, P, 2)
    aw = torch.softmax(proj[..., M * L * P * 2:].view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    vr = torch.tensor([[wv / w, hv / h] for (h, w), (hv, wv) in zip(shapes, valid)], dtype=torch.float32)   # (w, h) per level
    if ref_dim == 2:
        refl = ref[:, :, None, :] * vr[None, None]
        norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
        loc = refl[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        refl = ref[:, :, None, :] * torch.cat([vr, vr], -1)[None, None]
        loc = refl[:, :, None, :, None, :2] + off / P * refl[:, :, None, :, None, 2:] * 0.5
    pad = torch.cat([torch.ones(h, w, dtype=torch.bool).index_put_(
        (torch.arange(hv)[:, None], torch.arange(wv)[None, :]), torch.tensor(False)).reshape(-1)
        for (h, w), (hv, wv) in zip(shapes, valid)])
    expect = O.msda_core(value.masked_fill(pad[None, :, None, None], 0.0), shapes, loc, aw)
    out = ops.msda_fused(dev(value), dev(proj), dev(ref), shapes, N, S, M, Lq, L, P, ref_dim, True, valid_hw=valid)
    close(out.view(N, Lq, M * 32), expect, 1e-4, 1e-4)


@pytest.mark.parametrize("h,w,hv,wv", [(9, 13, 7, 10), (12, 20, 12, 17), (6, 10, 5, 10)])
def test_pos_sine2d_padded(ops, h, w, hv, wv):
    """Position map of a padded grid (position_encoding.py:64-84 with a mask): exact on the valid region.  At padded positions
    the reference's embedding is (0 - 0.5) / 1e-6 * 2 pi = -3.1e6, whose sin / cos depend on the last bit of pow(10000, .)
    -- no two implementations agree there -- so there the test checks the ARGUMENT the kernel uses (via the channel whose
    divisor is 1) and that the values are sines / cosines of something (bounded, sin^2 + cos^2 = 1 per channel pair)."""
    mask = torch.ones(2, h, w, dtype=torch.bool)
    mask[:, :hv, :wv] = False
    ref = O.pos_sine_2d(mask, 128).permute(0, 2, 3, 1)       # [2, h, w, 256]
    out = ops.pos_sine2d(2, h, w, 128, "cuda", valid=(hv, wv)).view(2, h, w, 256).cpu()
    close(out[:, :hv, :wv], ref[:, :hv, :wv], 1e-5, 1e-5)
    assert torch.isfinite(out).all() and out.abs().max() <= 1.0 + 1e-6
    pairs = out.view(2, h, w, 128, 2)
    close(pairs.pow(2).sum(-1), torch.ones(2, h, w, 128), 1e-5, 1e-5)
    # channels 0 / 1 (divisor 10000^0 = 1): sin / cos of the embedding itself -- equal to the reference's wherever the embedding is
    # small (valid rows of padded columns: y-embedding 0 there means (0 - 0.5) / 1e-6, but the x-embedding of a padded ROW ...)
    if wv < w:   # padded columns, valid rows: the x channels (128..) carry min(x + 1, wv) = wv, an ordinary angle
        close(out[:, :hv, wv:, 128:], ref[:, :hv, wv:, 128:], 1e-5, 1e-5)
    if hv < h:   # padded rows, valid columns: the y channels carry hv
        close(out[:, hv:, :wv, :128], ref[:, hv:, :wv, :128], 1e-5, 1e-5)
