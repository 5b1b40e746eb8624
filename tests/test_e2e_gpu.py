"""GPU: the full HIP per-clip forward (through build_model / forward) against
  (a) outputs of the REFERENCE itself (tests/golden/e2e_*.npz) and
  (b) the CPU oracle on the same seeded inputs and weights.
Tolerance: the north star asks for masks within 1e-3 IoU of the reference; we additionally bound the
max-abs logit error (fp32 everywhere; differences come only from summation order)."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tce_oracle as O  # noqa: E402
from _util import load_npz, synth_frames  # noqa: E402


def _args(backbone):
    return argparse.Namespace(backbone=backbone, with_box_refine=True, binary=True, freeze_text_encoder=True, f_token=8,
                              qtrans=True, num_feature_levels=4, text_encoder_layers=1)


@pytest.fixture(scope="module")
def models():
    cache = {}

    def get(backbone, salt):
        from tce_rvos_amd import build_model, load_synth_weights
        key = backbone
        if key not in cache:
            m, _, _ = build_model(_args(backbone))
            cache[key] = m.cuda().eval()
        m = cache[key]
        load_synth_weights(m, salt)
        m.repack()
        return m
    return get


def _compare(out, fx, atol_mask, prefix="out_"):
    res = {}
    for k in ("pred_logits", "pred_boxes", "reference_points", "pred_masks"):
        ref = torch.from_numpy(fx[prefix + k])
        res[k] = (out[k].cpu() - ref).abs().max().item()
    print("max abs diffs vs reference:", res)
    assert res["pred_logits"] < 2e-3 and res["pred_boxes"] < 1e-4 and res["reference_points"] < 1e-4
    assert res["pred_masks"] < atol_mask
    a = out["pred_masks"].cpu()[0] > 0
    b = torch.from_numpy(fx[prefix + "pred_masks"])[0] > 0
    iou = O.mask_iou(a, b)
    print("all-query mask IoU vs reference:", iou)
    assert iou > 1 - 1e-3
    mem = out["memory"].double().abs().sum().item()
    assert abs(mem - float(fx["out_memory_abs_sum"])) / float(fx["out_memory_abs_sum"]) < 1e-5


def _run(models, fixture, backbone):
    fx = load_npz(fixture)
    T, H, W = (int(v) for v in fx["thw"])
    model = models(backbone, int(fx["weights_salt"]))
    frames = synth_frames(T, H, W, int(fx["frames_seed"])).cuda()
    out = model.forward_features(frames, torch.from_numpy(fx["text_hidden"])[0].cuda(),
                                 torch.from_numpy(fx["text_pooled"])[0].cuda(), float(H), float(W))
    torch.cuda.synchronize()
    return fx, out, model


def test_swin_t_small_matches_reference(models):
    fx, out, _ = _run(models, "e2e_swin_t_small.npz", "swin_t_p4w7")
    _compare(out, fx, 5e-3)
    assert (out["memory"].cpu() - torch.from_numpy(fx["out_memory"])).abs().max().item() < 1e-3
    for i in range(3):
        d = (out["aux_outputs"][i]["pred_masks"].cpu() - torch.from_numpy(fx[f"aux{i}_pred_masks"])).abs().max().item()
        assert d < 5e-3, (i, d)


def test_swin_t_config2_fullsize_matches_reference(models):
    """BASELINE config 2: T=5, 360x640, Swin-T."""
    fx, out, _ = _run(models, "e2e_swin_t_cfg2.npz", "swin_t_p4w7")
    _compare(out, fx, 2e-2)


def test_resnet50_small_matches_reference(models):
    """Row A11 (ResNet-50 + FrozenBatchNorm2d backbone) against outputs of the reference itself."""
    fx, out, _ = _run(models, "e2e_resnet50_small.npz", "resnet50")
    _compare(out, fx, 5e-3)
    assert (out["memory"].cpu() - torch.from_numpy(fx["stage_memory"])).abs().max().item() < 2e-3


def test_resnet50_config1_fullsize_matches_reference(models):
    """BASELINE config 1: ResNet-50, T=1, 360x640 -- through the forward boundary, graph replay included."""
    fx, out, model = _run(models, "e2e_resnet50_cfg1.npz", "resnet50")
    _compare(out, fx, 2e-2)
    T, H, W = (int(v) for v in fx["thw"])
    frames = synth_frames(T, H, W, int(fx["frames_seed"])).cuda()
    ids = torch.randint(3, 50000, (1, 9))
    a = model([frames], ids, [{"size": torch.tensor([H, W])}])
    b = model([frames], ids, [{"size": torch.tensor([H, W])}])
    torch.cuda.synchronize()
    assert torch.equal(a["pred_masks"], b["pred_masks"])
    assert a["pred_masks"].shape == (1, 1, 5, 90, 160)


def test_forward_boundary_matches_oracle_and_is_deterministic(models):
    """Through model(samples, captions, targets) with token ids; second call must be bit-identical
    (the arena / cached constants must not leak state between clips)."""
    model = models("swin_t_p4w7", 7)
    T, H, W = 4, 96, 132
    frames = synth_frames(T, H, W, 99)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, 50000, (1, 11), generator=g)
    ids[0, 0], ids[0, -1] = 0, 2
    tgt = [{"size": torch.tensor([H, W])}]
    out1 = model([frames.cuda()], ids, tgt)
    other = model([synth_frames(2, 64, 64, 3).cuda()], ids, [{"size": torch.tensor([64, 64])}])  # different shape in between
    out2 = model([frames.cuda()], ids, tgt)
    torch.cuda.synchronize()
    for k in ("pred_logits", "pred_boxes", "pred_masks", "memory"):
        assert torch.equal(out1[k], out2[k]), k
    hid, pooled = model.forward_text_encoder(ids, "cuda")
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ref = O.forward(sd, O.OracleConfig(), frames, hid.cpu(), pooled.cpu(), img_size=(H, W))
    for k, tol in (("pred_logits", 2e-3), ("pred_boxes", 1e-4), ("pred_masks", 5e-3), ("reference_points", 1e-4)):
        d = (out1[k].cpu() - ref[k]).abs().max().item()
        assert d < tol, (k, d)
    assert O.mask_iou(out1["pred_masks"].cpu() > 0, ref["pred_masks"] > 0) > 1 - 1e-3
    # caller harness H on both sides (inference_ytvos.py:238-250)
    ma, qa = O.select_masks(out1["pred_logits"].cpu()[0], out1["pred_masks"].cpu()[0], (H, W))
    mb, qb = O.select_masks(ref["pred_logits"][0], ref["pred_masks"][0], (H, W))
    assert qa == qb and O.mask_iou(ma, mb) > 1 - 1e-3
    assert set(out1) == {"pred_logits", "pred_boxes", "pred_masks", "aux_outputs", "reference_points", "memory"}
    assert tuple(out1["pred_masks"].shape) == (1, T, 5, 24, 33) and tuple(out1["memory"].shape)[0] == T


def test_video_swin_t_small_matches_reference(models):
    """Video-Swin-T backbone, T=9 (> window depth 8: temporal padding + temporal shift path)."""
    fx, out, _ = _run(models, "e2e_vswin_t_small.npz", "video_swin_t_p4w7")
    _compare(out, fx, 5e-3)


@pytest.mark.parametrize("backbone,T,H,W", [("video_swin_t_p4w7", 8, 384, 640),   # BASELINE config 3
                                            ("swin_b_p4w7", 10, 480, 854),       # BASELINE config 5 (fp16 MFMA, f32 acc)
                                            # long clips (DAVIS clip_size 32, inference_davis.py:209-213):
                                            # FTF token sequence T*8 = 256, IQT over 32 frames, MSDA with N = 32
                                            ("swin_t_p4w7", 32, 120, 216),
                                            # Video-Swin beyond one temporal window: T=19 pads to 24, temporal shift 4
                                            ("video_swin_t_p4w7", 19, 96, 160)])
def test_fullsize_configs_match_oracle(models, backbone, T, H, W):
    """BASELINE configs 3 and 5 at full size: HIP path vs the CPU oracle (itself pinned to the reference by the
    golden fixtures) on the same synthetic weights / inputs."""
    torch.set_num_threads(min(32, torch.get_num_threads()))
    model = models(backbone, 11)
    frames = synth_frames(T, H, W, 123)
    g = torch.Generator().manual_seed(7)
    hid = torch.randn(32, 768, generator=g)
    pooled = torch.tanh(torch.randn(768, generator=g))
    out = model.forward_features(frames.cuda(), hid.cuda(), pooled.cuda(), float(H), float(W))
    torch.cuda.synchronize()
    sd = {k: v.cpu() for k, v in model.state_dict().items() if not k.startswith("text_encoder.")}
    b = __import__("tce_rvos_amd.config", fromlist=["BACKBONES"]).BACKBONES[backbone]
    cfg = O.OracleConfig(backbone=backbone, embed_dim=b["embed_dim"], depths=b["depths"], num_heads=b["num_heads"])
    # same clip through the public boundary (graph capture + replay at this shape)
    ids = torch.randint(3, 50000, (1, 12), generator=g)
    via_boundary = model([frames.cuda()], ids, [{"size": torch.tensor([H, W])}])
    assert via_boundary["pred_masks"].shape == (1, T, 5, (H + 3) // 4, (W + 3) // 4)
    with torch.no_grad():
        ref = O.forward(sd, cfg, frames, hid[None], pooled[None], img_size=(H, W))
    diffs = {k: (out[k].cpu() - ref[k]).abs().max().item() for k in ("pred_logits", "pred_boxes", "pred_masks")}
    iou = O.mask_iou(out["pred_masks"].cpu() > 0, ref["pred_masks"] > 0)
    print(backbone, "max abs diffs vs oracle:", diffs, "IoU", iou)
    assert diffs["pred_logits"] < 5e-3 and diffs["pred_boxes"] < 2e-4 and diffs["pred_masks"] < 5e-2
    assert iou > 1 - 1e-3


@pytest.mark.parametrize("flags", [dict(with_box_refine=False, qtrans=True, f_token=8),
                                   dict(with_box_refine=True, qtrans=False, f_token=0),
                                   dict(with_box_refine=False, qtrans=False, f_token=3, aux_loss=False)])
def test_flag_combinations_match_oracle(flags):
    """The reference's model-variation flags (opts.py:25,78,148,153): box refinement on/off, IQT on/off, frame tokens
    0 / 3 / 8, aux outputs on/off -- HIP path vs oracle on the same weights."""
    from tce_rvos_amd import build_model
    ns = _args("swin_t_p4w7")
    for k, v in flags.items():
        setattr(ns, k, v)
    model, _, _ = build_model(ns)
    model = model.cuda().eval()
    T, H, W = 3, 80, 120
    frames = synth_frames(T, H, W, 5)
    g = torch.Generator().manual_seed(3)
    hid, pooled = torch.randn(9, 768, generator=g), torch.tanh(torch.randn(768, generator=g))
    out = model.forward_features(frames.cuda(), hid.cuda(), pooled.cuda(), float(H), float(W))
    torch.cuda.synchronize()
    sd = {k: v.cpu() for k, v in model.state_dict().items() if not k.startswith("text_encoder.")}
    cfg = O.OracleConfig(with_box_refine=flags.get("with_box_refine", True), qtrans=flags.get("qtrans", True),
                         f_token=flags.get("f_token", 8), aux_loss=flags.get("aux_loss", True))
    with torch.no_grad():
        ref = O.forward(sd, cfg, frames, hid[None], pooled[None], img_size=(H, W))
    for k, tol in (("pred_logits", 2e-3), ("pred_boxes", 1e-4), ("pred_masks", 5e-3), ("reference_points", 1e-4)):
        d = (out[k].cpu() - ref[k]).abs().max().item()
        assert d < tol, (flags, k, d)
    assert ("aux_outputs" in out) == flags.get("aux_loss", True)
    if "aux_outputs" in out:
        for a, b in zip(out["aux_outputs"], ref["aux_outputs"]):
            assert (a["pred_boxes"].cpu() - b["pred_boxes"]).abs().max().item() < 1e-4
            assert (a["pred_masks"].cpu() - b["pred_masks"]).abs().max().item() < 5e-3


@pytest.mark.parametrize("layers,L", [(1, 7), (12, 32), (2, 100)])
def test_text_encoder_hip_matches_huggingface(layers, L):
    """RoBERTa on the HIP kernels vs HuggingFace's own forward of the same module (same weights, same ids)."""
    import transformers
    from tce_rvos_amd.text_encoder import TextPlan
    torch.manual_seed(layers)
    hf = transformers.RobertaModel(transformers.RobertaConfig(
        vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, num_hidden_layers=layers)).cuda().eval()
    g = torch.Generator().manual_seed(L)
    ids = torch.randint(3, 50264, (1, L), generator=g)
    ids[0, 0], ids[0, -1] = 0, 2
    ids = ids.cuda()
    with torch.no_grad():
        enc = hf(input_ids=ids, attention_mask=torch.ones_like(ids))
        hid, pooled = TextPlan(hf).forward(ids, lambda *s: torch.empty(*s, dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    d1 = (hid - enc.last_hidden_state[0]).abs().max().item()
    d2 = (pooled - enc.pooler_output[0]).abs().max().item()
    print("text encoder max abs diff: hidden", d1, "pooled", d2)
    assert d1 < 2e-4 and d2 < 1e-4
