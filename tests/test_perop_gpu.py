"""GPU: the HIP kernels against inputs / outputs of the REFERENCE's own sub-modules (tests/golden/perop_*.npz; SURVEY 8c list
(ii)-(vii)).  Two forms: (A) the reference module's INPUT goes straight through the HIP ops that replace the
module (Swin block, window attention, patch merging, fusion module, mask head, 3-D window attention); (B) for the modules whose
HIP counterpart lives inside the launch program (VisionLanguageBlocks, FrameTokenLayer, encoder / decoder layers) the clip of the
fixture runs through the whole HIP forward with taps on and every tap is compared with the reference module's OUTPUT."""
import argparse
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tce_oracle as O  # noqa: E402
from _util import load_npz, synth_frames, synth_sd_from_manifest  # noqa: E402

B = "backbone.0.body."


def _t(fx, key):
    return torch.from_numpy(np.ascontiguousarray(fx[key]))


def dev(t):
    return t.cuda().contiguous()


def _rel(got, ref):
    return float((got.detach().cpu().double() - ref.double()).abs().max() / ref.double().abs().max())


@pytest.fixture(scope="module")
def ops():
    from tce_rvos_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def swin():
    fx = load_npz("perop_swin_t.npz")
    return fx, synth_sd_from_manifest("statedict_swin_t.json", int(fx["weights_salt"]))


@pytest.mark.parametrize("layer,block,nh,shift", [(0, 1, 3, 3), (1, 0, 6, 0)])
def test_swin_block_fused_kernels_match_reference_module(ops, swin, layer, block, nh, shift):
    """SwinTransformerBlock (swin_transformer.py:202-258) = swin_attn_fused_kernel + ffn_fused_kernel<C, GELU>."""
    fx, sd = swin
    tag, p = f"backbone_0_body_layers_{layer}_blocks_{block}", f"{B}layers.{layer}.blocks.{block}."
    H, W = int(fx[tag + "_attr_H"]), int(fx[tag + "_attr_W"])
    x = _t(fx, tag + "_in0")
    T, _, C = x.shape
    d = {k[len(p):]: dev(v) for k, v in sd.items() if k.startswith(p)}
    xd = dev(x.reshape(T * H * W, C))
    pk = ops.swin_attn_pack(d["attn.qkv.weight"], d["attn.proj.weight"])
    ops.swin_attn_fused(xd, pk, d["attn.qkv.bias"], d["attn.proj.bias"], d["attn.relative_position_bias_table"], d["norm1.weight"],
                        d["norm1.bias"], T, H, W, C, shift)
    ffn = ops.ffn_pack(d["mlp.fc1.weight"], d["mlp.fc1.bias"], d["mlp.fc2.weight"])
    ops.ffn_fused(xd, ffn, d["mlp.fc2.bias"], 4 * C, ops.ACT_GELU, ln_in=(d["norm2.weight"], d["norm2.bias"]))
    ref = _t(fx, tag + "_out").reshape(T * H * W, C)
    assert _rel(xd, ref) < 2e-5


def test_window_attention_kernels_match_reference_module(ops, swin):
    """WindowAttention (swin_transformer.py:127-158) with the reference's own -100 mask: the module's input windows are un-
    partitioned back to tokens (pure data movement), the HIP three-launch form (qkv GEMM, window kernel, proj GEMM) runs on them
    and is compared with the module's output brought back to token order the same way."""
    fx, sd = swin
    tag, p = "backbone_0_body_layers_0_blocks_1_attn", B + "layers.0.blocks.1.attn."
    T, H, W, C, nh, ws, shift = 3, 18, 25, 96, 3, 7, 3
    Hp, Wp = 21, 28

    def to_tokens(win):  # [nW*T, 49, C] -> [T*H*W, C]: window_reverse, roll back, crop (swin_transformer.py:241-249)
        y = win.view(T, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(T, Hp, Wp, C)
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
        return y[:, :H, :W].reshape(T * H * W, C)

    xn = dev(to_tokens(_t(fx, tag + "_in0")))   # = norm1(x) at the real tokens
    d = {k[len(p):]: dev(v) for k, v in sd.items() if k.startswith(p)}
    qkv = ops.gemm(xn, d["qkv.weight"], bias=d["qkv.bias"])
    ref = to_tokens(_t(fx, tag + "_out"))
    from tce_rvos_amd._lib import lib
    for form in (1, 0, 2):  # split-fp16 matrix-core kernel, VALU kernel, exact-fp32 matrix-core kernel
        lib().tce_debug_window_attn_set_mfma(form)
        try:
            att = ops.window_attn(qkv, d["qkv.bias"], d["relative_position_bias_table"], T, H, W, C, nh, shift)
        finally:
            lib().tce_debug_window_attn_set_mfma(1)
        out = ops.gemm(att, d["proj.weight"], bias=d["proj.bias"])
        assert _rel(out, ref) < 2e-5, form


def test_patch_merging_matches_reference_module(ops, swin):
    """PatchMerging at 9 x 13 (swin_transformer.py:273-299) = patch_merge_ln_kernel + the reduction GEMM."""
    fx, sd = swin
    tag, p = "backbone_0_body_layers_1_downsample", B + "layers.1.downsample."
    x = _t(fx, tag + "_in0")
    T, _, C = x.shape
    H, W = int(fx[tag + "_in1"]), int(fx[tag + "_in2"])
    xm, H2, W2 = ops.patch_merge_ln(dev(x.reshape(-1, C)), dev(sd[p + "norm.weight"]), dev(sd[p + "norm.bias"]), T, H, W, C)
    out = ops.gemm(xm, dev(sd[p + "reduction.weight"]))
    ref = _t(fx, tag + "_out")
    assert (H2, W2) == (5, 7) and _rel(out, ref.reshape(-1, 2 * C)) < 2e-5


def test_fusion_module_matches_reference_module(ops, swin):
    """VisionLanguageFusionModule (segmentation.py:455-464) = few-row k / v projections + the folded cross-attention launch
    (ffn_fused_kernel<256,4,softmax>, multiply residual)."""
    fx, sd = swin
    p = "fusion_module.multihead_attn."
    tgt, text, pos = _t(fx, "fusion_module_kw_tgt")[:, 0], _t(fx, "fusion_module_kw_memory")[:, 0], _t(fx, "fusion_module_kw_pos")[:, 0]
    Wi, Bi = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    L, D = text.shape
    k = ops.gemm(dev(text), dev(Wi[D:2 * D]), bias=dev(Bi[D:2 * D]), a2=dev(pos))
    v = ops.gemm(dev(text), dev(Wi[2 * D:]), bias=dev(Bi[2 * D:]))
    wqT = ops.xattn_static(dev(Wi[:D]), dev(Bi[:D]))
    bufs = []

    def alloc(*shape, dtype=torch.float32):
        bufs.append(torch.empty(*shape, dtype=dtype, device="cuda"))
        return bufs[-1]

    pk = ops.xattn_pack(k, v, wqT, dev(sd[p + "out_proj.weight"]), L, alloc)
    x = dev(tgt)
    out = torch.empty_like(x)
    ops.xattn_fused(x, pk, dev(sd[p + "out_proj.bias"]), x.shape[0], out, res_mode=ops.RES_MUL)
    assert _rel(out, _t(fx, "fusion_module_out")[:, 0]) < 2e-5


def test_dynamic_mask_head_matches_reference_method(ops, swin):
    """dynamic_mask_with_coords (tce_rvos.py:426-490) = mask_pack_kernel + the first-layer GEMM + mask_tail_kernel."""
    fx, _ = swin
    feats, params, refs = _t(fx, "maskhead_in0")[0], _t(fx, "maskhead_in1")[0], _t(fx, "maskhead_in2")[0]
    T, Cm, h, w = feats.shape
    Q = params.shape[0] // T
    img = tuple(int(v) for v in fx["maskhead_size"])
    feats_cl = dev(feats.permute(0, 2, 3, 1).reshape(T, h * w, Cm))
    w0f = torch.empty(T, Q * 8, Cm, device="cuda")
    tail = torch.empty(1, T * Q, 112, device="cuda")
    ops.mask_pack(dev(params[None]), 1, T, Q, Cm, w0f, tail)
    G = torch.empty(T, h * w, Q * 8, device="cuda")
    ops.gemm_batched(feats_cl, w0f, G)
    masks = torch.empty(1, T, Q, h, w, device="cuda")
    ops.mask_tail(G, tail, dev(refs[None]), 2, masks, 1, T, Q, h, w, img[0], img[1])
    assert _rel(masks.view(T * Q, h, w), _t(fx, "maskhead_out")[0]) < 2e-5


@pytest.mark.parametrize("fixture,T", [("perop_vswin_t.npz", 9), ("perop_vswin_t_short.npz", 3)])
def test_window_attention_3d_kernels_match_reference_module(ops, fixture, T):
    """WindowAttention3D (video_swin_transformer.py:138-169) in a shifted block: T = 9 -> (8,7,7) windows (two temporal windows,
    region mask), T = 3 -> (3,7,7) windows with the [:N,:N] slice of the relative-position table."""
    fx = load_npz(fixture)
    sd = synth_sd_from_manifest("statedict_vswin_t.json", int(fx["weights_salt"]))
    tag, p = "backbone_0_body_layers_0_blocks_1_attn", B + "layers.0.blocks.1.attn."
    H, W, C, nh = 18, 25, 96, 3
    ws, ss = O.get_window_size_3d((T, H, W), (8, 7, 7), (4, 3, 3))
    Dp, Hp, Wp = -(-T // ws[0]) * ws[0], 21, 28

    def to_tokens(win):  # window_reverse + roll back + crop (video_swin_transformer.py:232-247)
        y = O.window_reverse_3d(win, ws, 1, Dp, Hp, Wp)
        if any(i > 0 for i in ss):
            y = torch.roll(y, shifts=(ss[0], ss[1], ss[2]), dims=(1, 2, 3))
        return y[:, :T, :H, :W].reshape(T * H * W, C)

    xn = dev(to_tokens(_t(fx, tag + "_in0")))
    d = {k[len(p):]: dev(v) for k, v in sd.items() if k.startswith(p)}
    qkv = ops.gemm(xn, d["qkv.weight"], bias=d["qkv.bias"])
    ref = to_tokens(_t(fx, tag + "_out"))
    from tce_rvos_amd._lib import lib
    for form in (1, 0):  # matrix-core kernel (split fp16), VALU kernel (exact fp32)
        lib().tce_debug_window_attn_set_mfma(form)
        try:
            att = ops.window_attn3d(qkv, d["qkv.bias"], d["relative_position_bias_table"], T, H, W, C, nh, True)
        finally:
            lib().tce_debug_window_attn_set_mfma(1)
        out = ops.gemm(att, d["proj.weight"], bias=d["proj.bias"])
        assert _rel(out, ref) < 2e-5, form


def _args(qtrans=True):
    return argparse.Namespace(backbone="swin_t_p4w7", with_box_refine=True, binary=True, freeze_text_encoder=True, f_token=8,
                              qtrans=qtrans, num_feature_levels=4, text_encoder_layers=1)


@pytest.mark.parametrize("fixture,qtrans", [("perop_swin_t.npz", True), ("perop_noqtrans.npz", False)])
def test_launch_program_taps_match_reference_modules(fixture, qtrans):
    """(B) the fixture's clip through the whole HIP forward (eager, taps on): the output of every VisionLanguageBlock (sr 8 / 4 /
    2 / 1), of the first FrameTokenLayer, of the first encoder layer and of decoder layers 0 / 1 (2-d / 4-d reference points; IQT
    on and off) against what the REFERENCE's module returned at that boundary.  Inputs of a module here are the HIP upstream's
    (fp32-class), so the bound is the end-to-end one: 2e-5 of the tensor's range per tap (5e-5 at the decoder: ~40 layers deep)."""
    from tce_rvos_amd import build_model, load_synth_weights, pipeline
    fx = load_npz(fixture)
    T, H, W = (int(v) for v in fx["thw"])
    model, _, _ = build_model(_args(qtrans))
    model = model.cuda().eval()
    load_synth_weights(model, int(fx["weights_salt"]))
    model.repack()
    model.use_graph = False
    frames = synth_frames(T, H, W, int(fx["frames_seed"])).cuda()
    old = pipeline.TAPS
    pipeline.TAPS = True
    try:
        out = model.forward_features(frames, _t(fx, "text_hidden")[0].cuda(), _t(fx, "text_pooled")[0].cuda(), float(H), float(W))
        torch.cuda.synchronize()
    finally:
        pipeline.TAPS = old
    taps = out["taps"][0]
    assert _rel(out["pred_masks"], _t(fx, "out_pred_masks")) < 2e-5
    checks = [("dec.hs", None)]
    if qtrans:
        checks += [(f"vl{s}", f"pixel_decoder_cross_attn_{s}_out") for s in (1, 2, 3, 4)]
        checks += [("L0.src4", "transformer_encoder_layers_0_ftoken_layers_out"), ("L0.src6", "transformer_encoder_layers_0_out")]
    for name, key in checks:
        if name == "dec.hs":
            for lid in (0, 1):
                ref = _t(fx, f"transformer_decoder_layers_{lid}_out")          # [T, Q, 256]
                assert _rel(taps[name][lid].reshape(ref.shape), ref) < 5e-5, (name, lid)
            continue
        ref = _t(fx, key)
        got = taps[name].reshape(-1, 256)
        assert _rel(got, ref.reshape(-1, 256)) < 2e-5, name
