"""CPU: the oracle (oracle/tce_oracle.py) against outputs of the reference itself (tests/golden/*.npz,
written by tests/golden/make_golden.py in the build container)."""
import numpy as np
import pytest
import torch

from oracle import tce_oracle as O
from _util import load_npz, synth_sd_from_manifest, synth_frames


def test_msda_core_matches_reference_cases():
    fx = load_npz("msda_cases.npz")
    for i in range(int(fx["n_cases"])):
        shapes = [tuple(int(v) for v in r) for r in fx[f"c{i}_shapes"]]
        out = O.msda_core(torch.from_numpy(fx[f"c{i}_value"]), shapes, torch.from_numpy(fx[f"c{i}_loc"]),
                          torch.from_numpy(fx[f"c{i}_w"]))
        ref = torch.from_numpy(fx[f"c{i}_out"])
        # the reference's own float tolerance is rtol 1e-2 / atol 1e-3 (models/ops/test.py:56); we are far inside it
        assert torch.allclose(out, ref, rtol=1e-4, atol=1e-6), (i, (out - ref).abs().max())


def test_msda_core_backward_matches_reference_gradients():
    """The oracle's statement of the native backward (ms_deform_attn_cuda.cu:105-186) against gradients of the
    reference's own core for a seeded grad_output (the comparison the reference's test makes for its CUDA op,
    models/ops/test.py:60-86)."""
    fx = load_npz("msda_cases.npz")
    for i in range(int(fx["n_cases"])):
        shapes = [tuple(int(v) for v in r) for r in fx[f"c{i}_shapes"]]
        gv, gl, gw = O.msda_core_backward(torch.from_numpy(fx[f"c{i}_value"]), shapes, torch.from_numpy(fx[f"c{i}_loc"]),
                                          torch.from_numpy(fx[f"c{i}_w"]), torch.from_numpy(fx[f"c{i}_gout"]))
        for got, key in ((gv, "gvalue"), (gl, "gloc"), (gw, "gw")):
            ref = torch.from_numpy(fx[f"c{i}_{key}"])
            assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5 * max(1.0, ref.abs().max().item())), (i, key, (got - ref).abs().max())


def test_interpolation_restatements_match_pytorch():
    fx = load_npz("interp_cases.npz")
    for i in range(int(fx["n_pairs"])):
        x = torch.from_numpy(fx[f"p{i}_in"])
        size = tuple(int(v) for v in fx[f"p{i}_size"])
        assert torch.equal(O.interp_nearest(x, size), torch.from_numpy(fx[f"p{i}_nearest"])), i
        b = O.interp_bilinear(x, size)
        assert torch.allclose(b, torch.from_numpy(fx[f"p{i}_bilinear"]), rtol=1e-5, atol=1e-5), i


def test_harness_matches_reference_caller():
    fx = load_npz("harness_cases.npz")
    for i in range(int(fx["n_cases"])):
        size = tuple(int(v) for v in fx[f"h{i}_size"])
        m, best = O.select_masks(torch.from_numpy(fx[f"h{i}_logits"])[0], torch.from_numpy(fx[f"h{i}_masks"])[0], size)
        assert best == int(fx[f"h{i}_best"])
        ref = torch.from_numpy(fx[f"h{i}_out"])
        assert O.mask_iou(m, ref) > 1 - 1e-4


def _run_e2e(fixture, manifest_name, backbone, **cfg_kw):
    fx = load_npz(fixture)
    T, H, W = (int(v) for v in fx["thw"])
    sd = synth_sd_from_manifest(manifest_name, int(fx["weights_salt"]))
    frames = synth_frames(T, H, W, int(fx["frames_seed"]))
    pad_mask = None
    if "pad_mask" in fx.files:  # a clip padded by the reference's nested_tensor_from_videos_list: zero frames + its mask
        Hp, Wp = (int(v) for v in fx["padded_hw"])
        padded = torch.zeros(T, 3, Hp, Wp)
        padded[:, :, :H, :W] = frames
        frames, pad_mask, (H, W) = padded, torch.from_numpy(fx["pad_mask"]), (Hp, Wp)
    cfg = O.OracleConfig(backbone=backbone, **cfg_kw)
    with torch.no_grad():
        out = O.forward(sd, cfg, frames, torch.from_numpy(fx["text_hidden"]), torch.from_numpy(fx["text_pooled"]),
                        img_size=(H, W), return_stages=True, pad_mask=pad_mask)
    return fx, out


def _check_outputs(fx, out, atol_mask):
    for k, atol in (("pred_logits", 1e-4), ("pred_boxes", 1e-5), ("reference_points", 1e-5), ("pred_masks", atol_mask)):
        ref = torch.from_numpy(fx["out_" + k])
        d = (out[k] - ref).abs().max().item()
        assert d < atol, (k, d)
    mem_sum = out["memory"].double().abs().sum().item()
    assert abs(mem_sum - float(fx["out_memory_abs_sum"])) / float(fx["out_memory_abs_sum"]) < 1e-5
    # thresholded-mask IoU of the selected query, the north-star metric
    t, q = out["pred_masks"].shape[1:3]
    a = out["pred_masks"][0] > 0
    b = torch.from_numpy(fx["out_pred_masks"])[0] > 0
    assert O.mask_iou(a, b) > 1 - 1e-3


def test_e2e_swin_t_small_matches_reference():
    fx, out = _run_e2e("e2e_swin_t_small.npz", "statedict_swin_t.json", "swin_t_p4w7")
    _check_outputs(fx, out, 2e-3)
    st = out["_stages"]
    for i in range(4):
        assert torch.allclose(st["backbone"][i], torch.from_numpy(fx[f"stage_backbone{i}"]), rtol=1e-4, atol=1e-4), i
    assert torch.allclose(st["memory"], torch.from_numpy(fx["stage_memory"]), rtol=1e-4, atol=1e-4)
    assert torch.allclose(st["mask_features"], torch.from_numpy(fx["stage_mask_features"]), rtol=1e-3, atol=1e-3)
    for i in range(3):
        assert torch.allclose(out["aux_outputs"][i]["pred_masks"], torch.from_numpy(fx[f"aux{i}_pred_masks"]),
                              rtol=1e-3, atol=2e-3)


def test_e2e_video_swin_t_small_matches_reference():
    fx, out = _run_e2e("e2e_vswin_t_small.npz", "statedict_vswin_t.json", "video_swin_t_p4w7")
    _check_outputs(fx, out, 2e-3)
    st = out["_stages"]
    for i in (1, 3):
        assert torch.allclose(st["backbone"][i], torch.from_numpy(fx[f"stage_backbone{i}"]), rtol=1e-4, atol=1e-4), i


def test_e2e_resnet50_small_matches_reference():
    """Row A11: reference FrozenBatchNorm2d / BackboneBase / Joiner over the restated ResNet-50 body."""
    fx, out = _run_e2e("e2e_resnet50_small.npz", "statedict_resnet50.json", "resnet50")
    _check_outputs(fx, out, 2e-3)
    st = out["_stages"]
    for i in (0, 3):
        assert torch.allclose(st["backbone"][i], torch.from_numpy(fx[f"stage_backbone{i}"]), rtol=1e-4, atol=1e-4), i


@pytest.mark.slow
def test_e2e_resnet50_config1_fullsize_matches_reference():
    """BASELINE config 1 (ResNet-50, T=1, 360x640)."""
    fx, out = _run_e2e("e2e_resnet50_cfg1.npz", "statedict_resnet50.json", "resnet50")
    _check_outputs(fx, out, 5e-3)


@pytest.mark.slow
def test_e2e_swin_t_config2_fullsize_matches_reference():
    """BASELINE config 2 (T=5, 360x640) -- ~15 s of CPU."""
    fx, out = _run_e2e("e2e_swin_t_cfg2.npz", "statedict_swin_t.json", "swin_t_p4w7")
    _check_outputs(fx, out, 5e-3)


def test_e2e_padded_clip_matches_reference():
    """The oracle's pad_mask branch (mask pyramid, masked cumulative position maps, valid ratios, zero-filled MSDA values,
    key-padding masks) against the REFERENCE run on the same padded clip: 3 x 90x140 through the reference's own
    nested_tensor_from_videos_list(size_divisibility=32) -> 96x160 (this branch was unpinned)."""
    fx, out = _run_e2e("e2e_swin_t_padded.npz", "statedict_swin_t.json", "swin_t_p4w7")
    assert tuple(fx["padded_hw"]) == (96, 160) and bool(fx["pad_mask"][:, 90:, :].all()) and not bool(fx["pad_mask"][:, :90, :140].any())
    _check_outputs(fx, out, 2e-3)
    st = out["_stages"]
    assert torch.allclose(st["memory"], torch.from_numpy(fx["stage_memory"]), rtol=1e-4, atol=1e-4)
    assert torch.allclose(st["mask_features"][:1], torch.from_numpy(fx["stage_mask_features"]), rtol=1e-3, atol=1e-3)
    for i in range(3):
        assert torch.allclose(out["aux_outputs"][i]["pred_masks"], torch.from_numpy(fx[f"aux{i}_pred_masks"]), rtol=1e-3, atol=2e-3)
    # the reference's own stride-8 position map (PositionEmbeddingSine2D, position_encoding.py:64-84): valid positions carry
    # angles <= 2*pi; FULLY PADDED columns / rows carry (0 - 0.5) / (0 + 1e-6) * 2*pi = -3.14e6 in the other axis' embedding,
    # i.e. sin / cos of -3.14e6 / 10000^(2i/128): a value whose low-frequency... rather HIGH-frequency channels depend on the
    # last bit of pow() (angle error 3e6 * 6e-8 = 0.2 rad) -- reproducible only by the same libm, which the oracle shares
    # with the reference here and the GPU does not (tests/test_e2e_gpu.py::test_padded_clip_matches_reference).
    pos = torch.from_numpy(fx["stage_pos1_frame0"])            # [256, 12, 20]; valid 12 x 18 (rows 90/96 -> all 12, cols 140/160)
    ours = O.pos_sine2d(torch.from_numpy(fx["pad_mask"])[:1, ::8, ::8])[0] if hasattr(O, "pos_sine2d") else None
    hv8 = int((~torch.from_numpy(fx["pad_mask"])[0, ::8, 0]).sum())
    wv8 = int((~torch.from_numpy(fx["pad_mask"])[0, 0, ::8]).sum())
    assert pos[:, :hv8, :wv8].abs().max() <= 1.0 + 1e-6
    if wv8 < 20:
        # y-embedding half (channels 0..127) at a fully padded column: the -3.14e6 angle
        col = pos[:128, 0, wv8]
        assert col.abs().max() <= 1.0 + 1e-6 and col.abs().min() < 0.999  # sin / cos of a huge angle, not a constant


def test_e2e_valid_indices_single_frame_path_matches_reference():
    """tce_rvos.py:233-243 (A2D / JHMDB: one annotated frame per clip): the reference run with targets[0]['valid_indices'] = 1
    on a 3-frame clip; every output has t = 1, `memory` is the selected frame's."""
    fx = load_npz("e2e_swin_t_valid_idx.npz")
    T, H, W = (int(v) for v in fx["thw"])
    sd = synth_sd_from_manifest("statedict_swin_t.json", int(fx["weights_salt"]))
    frames = synth_frames(T, H, W, int(fx["frames_seed"]))
    with torch.no_grad():
        out = O.forward(sd, O.OracleConfig(), frames, torch.from_numpy(fx["text_hidden"]), torch.from_numpy(fx["text_pooled"]),
                        img_size=(H, W), valid_index=int(fx["valid_index"]))
    assert tuple(out["pred_masks"].shape) == (1, 1, 5, 18, 25) == tuple(fx["out_pred_masks"].shape)
    for k, atol in (("pred_logits", 1e-4), ("pred_boxes", 1e-5), ("reference_points", 1e-5), ("pred_masks", 2e-3), ("memory", 1e-4)):
        ref = torch.from_numpy(fx["out_" + k])
        assert tuple(out[k].shape) == tuple(ref.shape), k
        assert (out[k] - ref).abs().max().item() < atol, k
    for i in range(3):
        assert torch.allclose(out["aux_outputs"][i]["pred_masks"], torch.from_numpy(fx[f"aux{i}_pred_masks"]), rtol=1e-3, atol=2e-3)
    # and it is NOT what the full clip gives for that frame (the frame-token / IQT stages mix the clip's frames)
    with torch.no_grad():
        full = O.forward(sd, O.OracleConfig(), frames, torch.from_numpy(fx["text_hidden"]), torch.from_numpy(fx["text_pooled"]), img_size=(H, W))
    assert (full["pred_masks"][:, 1:2] - out["pred_masks"]).abs().max().item() > 1e-2


def test_e2e_vis_loss_and_contrastive_outputs_match_reference():
    """The optional output keys of A27 (tce_rvos.py:360-365): pred_visible (--vis_loss: visible_embed heads, aux levels included)
    and contrastive (--contrastive: cosine of the frame-mean memory and the sentence feature) against the reference run with
    both flags."""
    fx = load_npz("e2e_swin_t_vis_contrastive.npz")
    T, H, W = (int(v) for v in fx["thw"])
    sd = synth_sd_from_manifest("statedict_swin_t_vis.json", int(fx["weights_salt"]))
    frames = synth_frames(T, H, W, int(fx["frames_seed"]))
    with torch.no_grad():
        out = O.forward(sd, O.OracleConfig(vis_loss=True, contrastive=True), frames, torch.from_numpy(fx["text_hidden"]),
                        torch.from_numpy(fx["text_pooled"]), img_size=(H, W))
    for k, atol in (("pred_logits", 1e-4), ("pred_boxes", 1e-5), ("pred_masks", 2e-3), ("pred_visible", 1e-4), ("contrastive", 1e-5)):
        ref = torch.from_numpy(fx["out_" + k])
        assert tuple(out[k].shape) == tuple(ref.shape), k
        assert (out[k] - ref).abs().max().item() < atol, k
    for i in range(3):
        assert torch.allclose(out["aux_outputs"][i]["pred_visible"], torch.from_numpy(fx[f"aux{i}_pred_visible"]), rtol=1e-4, atol=1e-4)


@pytest.mark.slow
def test_e2e_video_swin_t_config3_fullsize_matches_reference():
    """BASELINE config 3 (Video-Swin-T, T=8, 384x640) -- ~25 s of CPU."""
    fx, out = _run_e2e("e2e_vswin_t_cfg3.npz", "statedict_vswin_t.json", "video_swin_t_p4w7")
    assert tuple(out["pred_masks"].shape) == (1, 8, 5, 96, 160)
    _check_outputs(fx, out, 5e-3)


@pytest.mark.slow
def test_e2e_swin_b_config5_fullsize_matches_reference():
    """BASELINE config 5's shapes (Swin-B: embed 128, depths 2/2/18/2, heads 4/8/16/32; T=10, 480x854) in the reference's own
    fp32 -- ~60 s of CPU.  The first reference vector Swin-B has."""
    fx, out = _run_e2e("e2e_swin_b_cfg5.npz", "statedict_swin_b.json", "swin_b_p4w7", embed_dim=128, depths=(2, 2, 18, 2),
                       num_heads=(4, 8, 16, 32))
    assert tuple(out["pred_masks"].shape) == (1, 10, 5, 120, 214)
    _check_outputs(fx, out, 1e-2)


def test_msda_c_restatement_matches_reference_cases():
    """oracle/msda_ref.c (scalar C statement of the native kernel's rule) against the reference fixture --
    including the reference's own D=2 test case (models/ops/test.py:21-26)."""
    import ctypes
    from oracle.build_oracle import build
    lib = ctypes.CDLL(build())
    fx = load_npz("msda_cases.npz")
    for i in range(int(fx["n_cases"])):
        value = np.ascontiguousarray(fx[f"c{i}_value"])
        loc = np.ascontiguousarray(fx[f"c{i}_loc"])
        w = np.ascontiguousarray(fx[f"c{i}_w"])
        shapes = np.ascontiguousarray(fx[f"c{i}_shapes"])
        lsi = np.concatenate([[0], np.cumsum(shapes[:, 0] * shapes[:, 1])[:-1]]).astype(np.int64)
        N, S, M, D = value.shape
        _, Lq, _, L, P, _ = loc.shape
        out = np.zeros((N, Lq, M * D), dtype=np.float32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        lib.msda_ref_forward(p(value), p(shapes), p(lsi), p(loc), p(w), p(out), N, S, M, D, Lq, L, P)
        assert np.allclose(out, fx[f"c{i}_out"], rtol=1e-4, atol=1e-6), i
