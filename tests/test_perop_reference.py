"""CPU: the oracle's per-op functions against inputs / outputs of the REFERENCE's own sub-modules (SURVEY 8c list (ii)-(vii);
VERDICT r4 #7).  tests/golden/perop_*.npz were captured by forward hooks while the reference ran a small clip with the synthetic
weights (tests/golden/make_golden.py round5): every kernel family's checker is thereby pinned at its own boundary, not only
end to end.  The GPU counterparts (tests/test_kernels_gpu.py::test_perop_*) feed the same inputs to the HIP ops."""
import numpy as np
import pytest
import torch

from oracle import tce_oracle as O
from _util import load_npz, synth_sd_from_manifest

B = "backbone.0.body."


def _t(fx, key):
    return torch.from_numpy(np.ascontiguousarray(fx[key]))


def _close(got, ref, rtol=1e-4, atol=1e-4):
    d = (got - ref).abs().max().item()
    assert torch.allclose(got, ref, rtol=rtol, atol=atol * max(1.0, ref.abs().max().item())), d


@pytest.fixture(scope="module")
def swin():
    fx = load_npz("perop_swin_t.npz")
    return fx, synth_sd_from_manifest("statedict_swin_t.json", int(fx["weights_salt"]))


def test_window_attention_shifted_masked(swin):
    """swin_transformer.py:127-158 with the reference's own -100 mask (12 windows of the padded 21 x 28 grid)."""
    fx, sd = swin
    tag = "backbone_0_body_layers_0_blocks_1_attn"
    mask = _t(fx, tag + "_kw_mask")
    out = O.window_attention(sd, B + "layers.0.blocks.1.attn.", _t(fx, tag + "_in0"), 3, 7, mask)
    _close(out, _t(fx, tag + "_out"))
    # the oracle's own mask construction equals the reference's (swin_transformer.py:370-388)
    assert torch.equal(O.shift_attn_mask(21, 28, 7, 3), mask)


@pytest.mark.parametrize("layer,block,nh,shift", [(0, 1, 3, 3), (1, 0, 6, 0)])
def test_swin_block_shifted_padded(swin, layer, block, nh, shift):
    """swin_transformer.py:202-258: 18 x 25 and 9 x 13 grids (neither a multiple of 7: pad AFTER norm1), shifted and not."""
    fx, sd = swin
    tag = f"backbone_0_body_layers_{layer}_blocks_{block}"
    H, W = int(fx[tag + "_attr_H"]), int(fx[tag + "_attr_W"])
    out = O.swin_block(sd, f"{B}layers.{layer}.blocks.{block}.", _t(fx, tag + "_in0"), H, W, nh, 7, shift, _t(fx, tag + "_in1"))
    _close(out, _t(fx, tag + "_out"))


def test_patch_merging_odd(swin):
    """swin_transformer.py:273-299 at 9 x 13 (odd in both dimensions)."""
    fx, sd = swin
    tag = "backbone_0_body_layers_1_downsample"
    out = O.patch_merging(sd, B + "layers.1.downsample.", _t(fx, tag + "_in0"), int(fx[tag + "_in1"]), int(fx[tag + "_in2"]))
    _close(out, _t(fx, tag + "_out"))


def test_vision_language_fusion(swin):
    """segmentation.py:455-464: tgt * MHA(tgt, text + pos, text)."""
    fx, sd = swin
    out = O.fusion(sd, _t(fx, "fusion_module_kw_tgt"), _t(fx, "fusion_module_kw_memory"),
                   _t(fx, "fusion_module_kw_memory_key_padding_mask").bool(), _t(fx, "fusion_module_kw_pos"))
    _close(out, _t(fx, "fusion_module_out"))


@pytest.mark.parametrize("stage,sr", [(1, 8), (2, 4), (3, 2), (4, 1)])
def test_vision_language_block(swin, stage, sr):
    """segmentation.py:326-377 at the four spatial-reduction ratios (reduced grids 2x3, 2x3, 2x3 and the full 3x4)."""
    fx, sd = swin
    tag = f"pixel_decoder_cross_attn_{stage}"
    out = O.vl_block(sd, f"pixel_decoder.cross_attn_{stage}.", O.OracleConfig(), _t(fx, tag + "_kw_tgt"), _t(fx, tag + "_kw_memory"),
                     int(fx[tag + "_kw_t"]), int(fx[tag + "_kw_h"]), int(fx[tag + "_kw_w"]),
                     _t(fx, tag + "_kw_tgt_key_padding_mask").bool(), _t(fx, tag + "_kw_memory_key_padding_mask").bool(),
                     _t(fx, tag + "_kw_pos"), _t(fx, tag + "_kw_query_pos"), sr)
    _close(out, _t(fx, tag + "_out"))


def _shapes(fx, key):
    return [tuple(int(v) for v in r) for r in fx[key]]


def test_frame_token_layer(swin):
    """tce_deformable_transformer.py:443-493."""
    fx, sd = swin
    tag = "transformer_encoder_layers_0_ftoken_layers"
    src, _ = O.frame_token_layer(sd, "transformer.encoder.layers.0.ftoken_layers.", O.OracleConfig(), _t(fx, tag + "_in0"),
                                 _t(fx, tag + "_in1"), _t(fx, tag + "_in2"), _t(fx, tag + "_in3"), _shapes(fx, tag + "_in4"),
                                 _t(fx, tag + "_in6").bool(), _t(fx, tag + "_in7"))
    _close(src, _t(fx, tag + "_out"))


def test_encoder_layer(swin):
    """tce_deformable_transformer.py:535-553 (frame-token layer + MSDA self-attention + FFN)."""
    fx, sd = swin
    tag = "transformer_encoder_layers_0"
    out, _ = O.encoder_layer(sd, "transformer.encoder.layers.0.", O.OracleConfig(), _t(fx, tag + "_in0"), _t(fx, tag + "_in1"),
                             _t(fx, tag + "_in2"), _shapes(fx, tag + "_in3"), _t(fx, tag + "_in5"), _t(fx, tag + "_in6").bool(),
                             _t(fx, tag + "_in7"), _t(fx, tag + "_in8"))
    _close(out, _t(fx, tag + "_out"))


@pytest.mark.parametrize("fixture,qtrans", [("perop_swin_t.npz", True), ("perop_noqtrans.npz", False)])
@pytest.mark.parametrize("lid", [0, 1])
def test_decoder_layer(fixture, qtrans, lid):
    """tce_deformable_transformer.py:675-699 with 2-d (layer 0) and 4-d (layer 1) reference points, IQT on and off."""
    fx = load_npz(fixture)
    sd = synth_sd_from_manifest("statedict_swin_t.json", int(fx["weights_salt"]))
    tag = f"transformer_decoder_layers_{lid}"
    cfg = O.OracleConfig(qtrans=qtrans)
    out, _, _ = O.decoder_layer(sd, f"transformer.decoder.layers.{lid}.", cfg, _t(fx, tag + "_in0"), _t(fx, tag + "_in1"),
                                _t(fx, tag + "_in2"), _t(fx, tag + "_in3"), _shapes(fx, tag + "_in4"), _t(fx, tag + "_in6").bool())
    assert _t(fx, tag + "_in2").shape[-1] == (2 if lid == 0 else 4)
    _close(out, _t(fx, tag + "_out"))


def test_dynamic_mask_with_coords(swin):
    """tce_rvos.py:426-490 (+ compute_locations :586-599, parse_dynamic_params :536-559)."""
    fx, _ = swin
    feats, params, refs = _t(fx, "maskhead_in0")[0], _t(fx, "maskhead_in1")[0], _t(fx, "maskhead_in2")[0]
    out = O.dynamic_mask_head(O.OracleConfig(), feats, params, refs, tuple(int(v) for v in fx["maskhead_size"]))
    ref = _t(fx, "maskhead_out")[0]
    assert (out - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("fixture,nwin_tokens", [("perop_vswin_t.npz", 392), ("perop_vswin_t_short.npz", 147)])
def test_window_attention_3d(fixture, nwin_tokens):
    """video_swin_transformer.py:138-169: T = 9 -> (8,7,7) windows with the temporal shift mask; T = 3 -> (3,7,7) windows and
    the [:N,:N] slice of the (8,7,7) relative-position table."""
    fx = load_npz(fixture)
    sd = synth_sd_from_manifest("statedict_vswin_t.json", int(fx["weights_salt"]))
    tag = "backbone_0_body_layers_0_blocks_1_attn"
    x = _t(fx, tag + "_in0")
    assert x.shape[1] == nwin_tokens
    out = O.window_attention_3d(sd, B + "layers.0.blocks.1.attn.", x, 3, (8, 7, 7), _t(fx, tag + "_kw_mask"))
    _close(out, _t(fx, tag + "_out"))
