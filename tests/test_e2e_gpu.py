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
    # relative form of the same bound: the synthetic weights give mask logits of magnitude 1e2..1e3, where an absolute
    # tolerance says little -- require |d| <= 2e-5 * max|ref| as well (fp32 round-off through ~60 layers is ~3e-6)
    scale = float(np.abs(fx[prefix + "pred_masks"]).max())
    print("pred_masks max|ref| %.1f -> relative error %.2e" % (scale, res["pred_masks"] / scale))
    assert res["pred_masks"] <= 2e-5 * scale
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
    fx, out, model = _run(models, "e2e_swin_t_cfg2.npz", "swin_t_p4w7")
    _compare(out, fx, 2e-2)
    # the same clip again: now captured and replayed as a hipGraph with all its parallel branches (text, early input
    # projections, four lateral paths, decoder, frame-token forks) -- must reproduce the eager, single-stream result bit for
    # bit (a missing join between branches shows up here as a mismatch or as run-to-run differences)
    T, H, W = (int(v) for v in fx["thw"])
    frames = synth_frames(T, H, W, int(fx["frames_seed"])).cuda()
    hid, pooled = torch.from_numpy(fx["text_hidden"])[0].cuda(), torch.from_numpy(fx["text_pooled"])[0].cuda()
    n_graphs = len(model._graphs)
    for _ in range(3):
        again = model.forward_features(frames, hid, pooled, float(H), float(W))
        torch.cuda.synchronize()
        for k in ("pred_logits", "pred_boxes", "pred_masks", "memory", "reference_points"):
            assert torch.equal(again[k], out[k]), k
    assert len(model._graphs) == n_graphs + 1  # captured on the second sighting, replayed afterwards


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
                                            # the same chunk length at DAVIS resolution (480p): N*S = 273 k tokens
                                            ("swin_t_p4w7", 32, 480, 854),
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
    scale = ref["pred_masks"].abs().max().item()
    print(backbone, "max abs diffs vs oracle:", diffs, "IoU", iou, "mask logit scale", scale)
    assert diffs["pred_logits"] < 5e-3 and diffs["pred_boxes"] < 2e-4 and diffs["pred_masks"] < 5e-2
    assert diffs["pred_masks"] <= 5e-5 * scale   # relative bound (oracle and HIP path both carry fp32 round-off here)
    assert iou > 1 - 1e-3
    # the same clip captured and replayed (all parallel branches live): bit-identical to the eager, single-stream pass.
    # (Round 3: a kernel variant that was right eagerly and at kernel level made replays differ from run to run here.)
    for _ in range(3):
        again = model.forward_features(frames.cuda(), hid.cuda(), pooled.cuda(), float(H), float(W))
        torch.cuda.synchronize()
        for k in ("pred_logits", "pred_boxes", "pred_masks", "memory"):
            assert torch.equal(again[k], out[k]), k


@pytest.mark.parametrize("fixture,backbone", [("e2e_vswin_t_cfg3.npz", "video_swin_t_p4w7"),   # BASELINE config 3
                                              ("e2e_swin_b_cfg5.npz", "swin_b_p4w7")])         # BASELINE config 5 (shapes)
def test_fullsize_configs_match_reference(models, fixture, backbone):
    """BASELINE configs 3 and 5 at full size against outputs of the REFERENCE ITSELF (tests/golden/make_golden.py round4,
    generated in the build container): Video-Swin-T T=8 384x640 and Swin-B T=10 480x854, default (fp32-class) arithmetic.
    Round 3 compared these two with the oracle only; Swin-B had no reference vector at any size."""
    fx, out, model = _run(models, fixture, backbone)
    _compare(out, fx, 5e-2)
    T, H, W = (int(v) for v in fx["thw"])
    frames = synth_frames(T, H, W, int(fx["frames_seed"])).cuda()
    hid, pooled = torch.from_numpy(fx["text_hidden"])[0].cuda(), torch.from_numpy(fx["text_pooled"])[0].cuda()
    for _ in range(2):  # captured, replayed: bit-identical to the eager pass
        again = model.forward_features(frames, hid, pooled, float(H), float(W))
        torch.cuda.synchronize()
        for k in ("pred_logits", "pred_boxes", "pred_masks", "memory"):
            assert torch.equal(again[k], out[k]), k


def test_config5_mixed_policy_matches_reference(models):
    """The shipped config-5 arithmetic mix (single-pass fp16 downstream of the queries) against the reference's own fp32
    output at full size: the 1e-3 IoU criterion of north_star, measured against the reference rather than the oracle."""
    fx = load_npz("e2e_swin_b_cfg5.npz")
    model = models("swin_b_p4w7", int(fx["weights_salt"]))
    model.set_arith_policy("cfg5_mixed")
    try:
        T, H, W = (int(v) for v in fx["thw"])
        frames = synth_frames(T, H, W, int(fx["frames_seed"])).cuda()
        out = model.forward_features(frames, torch.from_numpy(fx["text_hidden"])[0].cuda(),
                                     torch.from_numpy(fx["text_pooled"])[0].cuda(), float(H), float(W))
        torch.cuda.synchronize()
        ref = torch.from_numpy(fx["out_pred_masks"])
        d = (out["pred_masks"].cpu() - ref).abs().max().item()
        iou = O.mask_iou(out["pred_masks"].cpu()[0] > 0, ref[0] > 0)
        print(f"cfg5_mixed vs reference: max|d| {d:.3e} of max|ref| {ref.abs().max().item():.1f}, IoU {iou:.6f}")
        assert iou > 1 - 1e-3 and d <= 5e-4 * ref.abs().max().item()
    finally:
        model.set_arith_policy("uniform")


def test_padded_clip_matches_reference(models):
    """A padded clip against the REFERENCE run on the same NestedTensor (its own nested_tensor_from_videos_list with
    size_divisibility=32: 3 x 90x140 -> 96x160 + pad mask).  Pins the padded path end to end, and shows the border-band
    claim of test_padded_clip_matches_oracle against the reference's own numbers: at fully padded columns the reference's
    position map is sin / cos of -3.14e6 / 10000^(2i/128) (stored in the fixture), which no other libm / device reproduces."""
    from tce_rvos_amd import nested_tensor_from_videos_list, ops
    fx = load_npz("e2e_swin_t_padded.npz")
    T, hv, wv = (int(v) for v in fx["thw"])
    H, W = (int(v) for v in fx["padded_hw"])
    model = models("swin_t_p4w7", int(fx["weights_salt"]))
    nt = nested_tensor_from_videos_list([synth_frames(T, hv, wv, int(fx["frames_seed"])).cuda()], size_divisibility=32)
    assert tuple(nt.tensors.shape[-2:]) == (H, W) and torch.equal(nt.mask[0].cpu(), torch.from_numpy(fx["pad_mask"]))
    out = model.forward_features(nt.tensors[0], torch.from_numpy(fx["text_hidden"])[0].cuda(),
                                 torch.from_numpy(fx["text_pooled"])[0].cuda(), float(H), float(W), valid_hw=(hv, wv))
    torch.cuda.synchronize()
    for k, tol in (("pred_logits", 2e-3), ("pred_boxes", 1e-4), ("reference_points", 1e-4)):
        d = (out[k].cpu() - torch.from_numpy(fx["out_" + k])).abs().max().item()
        assert d < tol, (k, d)
    pm, rm = out["pred_masks"].cpu(), torch.from_numpy(fx["out_pred_masks"])
    dm = (pm - rm).abs()[0]
    scale = rm.abs().max().item()
    hv4, wv4 = hv // 4, wv // 4
    band = {m: dm[..., :hv4 - m, :wv4 - m].max().item() for m in (0, 4, 8, 12)}
    print(f"padded clip vs REFERENCE: max|ref| {scale:.1f}; max|d| inside the valid region minus margin: {band}; "
          f"whole map {dm.max().item():.2e}")
    assert band[12] < 5e-3 + 2e-5 * scale and dm.max().item() < 2e-3 * scale
    inner = (slice(None), slice(None), slice(None), slice(0, hv4 - 8), slice(0, wv4 - 8))
    assert O.mask_iou(pm[inner] > 0, rm[inner] > 0) > 1 - 1e-3
    # memory (encoder output): valid tokens match tightly; the reference's own values at PADDED tokens are what differs
    mem, rmem = out["memory"].cpu(), torch.from_numpy(fx["out_memory"])
    lv = [(12, 20), (6, 10), (3, 5), (2, 3)]
    valid_tok = torch.cat([((torch.arange(h)[:, None] * (H // h) < hv) & (torch.arange(w)[None, :] * (W // w) < wv)).reshape(-1)
                           for h, w in lv[:3]] + [torch.zeros(6, dtype=torch.bool)])  # (level 3's mask is resampled differently)
    dmem = (mem - rmem).abs().amax(-1)
    print(f"memory: max|d| valid tokens {dmem[:, valid_tok].max().item():.2e}, padded tokens {dmem[:, ~valid_tok].max().item():.2e}")
    assert dmem[:, valid_tok].max().item() < 2e-3
    # the stride-8 position map: ours (tce_pos_sine2d_valid_f32) vs the reference's, valid region vs padded columns
    pos_ref = torch.from_numpy(fx["stage_pos1_frame0"]).permute(1, 2, 0)                    # [12, 20, 256]
    hv8 = int((~torch.from_numpy(fx["pad_mask"])[0, ::8, 0]).sum())
    wv8 = int((~torch.from_numpy(fx["pad_mask"])[0, 0, ::8]).sum())
    pos = ops.pos_sine2d(1, 12, 20, 128, "cuda", valid=(hv8, wv8)).cpu().view(12, 20, 256)
    d_valid = (pos[:hv8, :wv8] - pos_ref[:hv8, :wv8]).abs().max().item()
    d_pad = (pos[:, wv8:] - pos_ref[:, wv8:]).abs().max().item() if wv8 < 20 else 0.0
    print(f"stride-8 position map vs the reference's: valid region max|d| {d_valid:.2e}; fully padded columns {d_pad:.2e} "
          f"(angle -3.14e6 / 10000^(2i/128): fp32 pow decides it)")
    assert d_valid < 2e-5


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


@pytest.mark.parametrize("L", [33, 70])
def test_captions_longer_than_32_tokens_match_oracle(models, L):
    """More than 32 text tokens: the five text cross-attention sites leave the folded one-launch form (built for <= 32 keys,
    csrc/chain.hip) for q-projection + attention over L keys + output projection (segmentation.py:366-371, tce_rvos.py:283-291);
    eager launches and graph replays of that form against the oracle."""
    model = models("swin_t_p4w7", 31)
    T, H, W = 3, 80, 120
    frames = synth_frames(T, H, W, 6)
    g = torch.Generator().manual_seed(L)
    hid, pooled = torch.randn(L, 768, generator=g), torch.tanh(torch.randn(768, generator=g))
    outs = []
    for _ in range(4):  # eager sightings first, then the captured graph
        outs.append(model.forward_features(frames.cuda(), hid.cuda(), pooled.cuda(), float(H), float(W)))
    torch.cuda.synchronize()
    out = outs[0]
    sd = {k: v.cpu() for k, v in model.state_dict().items() if not k.startswith("text_encoder.")}
    with torch.no_grad():
        ref = O.forward(sd, O.OracleConfig(), frames, hid[None], pooled[None], img_size=(H, W))
    for k, tol in (("pred_logits", 2e-3), ("pred_boxes", 1e-4), ("pred_masks", 5e-3)):
        d = (out[k].cpu() - ref[k]).abs().max().item()
        assert d < tol, (L, k, d)
    for k in ("pred_logits", "pred_boxes", "pred_masks", "memory"):
        assert torch.equal(outs[-1][k], out[k]), k


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


def test_unit_scale_mask_logits_make_the_iou_criterion_bite(models):
    """The default synthetic weights give mask logits of magnitude ~1e2, so almost no pixel sits near the threshold
    and `IoU > 1 - 1e-3` would survive large errors.  Here the controller's last layer is scaled so that the dynamic
    mask head produces logits of O(1) (a realistic, trained-like regime): a visible share of pixels lies within 1e-2
    of the threshold, and the 1e-3 IoU criterion of the north star has to be earned."""
    model = models("swin_t_p4w7", 21)
    with torch.no_grad():
        for k in ("controller.layers.2.weight", "controller.layers.2.bias"):
            model.state_dict(keep_vars=True)[k].mul_(0.2)   # the mask MLP is cubic in its generated parameters
    model.repack()
    T, H, W = 5, 180, 320
    frames = synth_frames(T, H, W, 31)
    g = torch.Generator().manual_seed(8)
    hid, pooled = torch.randn(20, 768, generator=g), torch.tanh(torch.randn(768, generator=g))
    out = model.forward_features(frames.cuda(), hid.cuda(), pooled.cuda(), float(H), float(W))
    torch.cuda.synchronize()
    sd = {k: v.cpu() for k, v in model.state_dict().items() if not k.startswith("text_encoder.")}
    with torch.no_grad():
        ref = O.forward(sd, O.OracleConfig(), frames, hid[None], pooled[None], img_size=(H, W))
    rm = ref["pred_masks"]
    near = (rm.abs() < 1e-2).float().mean().item()
    d = (out["pred_masks"].cpu() - rm).abs().max().item()
    iou = O.mask_iou(out["pred_masks"].cpu() > 0, rm > 0)
    print(f"mask logits: mean|x| {rm.abs().mean():.3f} std {rm.std():.3f}; {100 * near:.2f} % of pixels within 1e-2 of the "
          f"threshold; max|d| {d:.2e}; IoU {iou:.6f}")
    assert 0.05 < rm.abs().mean().item() < 20.0, "salt no longer gives O(1) logits"
    assert near > 1e-3, "too few pixels near the threshold for the IoU criterion to discriminate"
    assert d <= 2e-5 * max(1.0, rm.abs().max().item()) + 2e-5
    assert iou > 1 - 1e-3
    load = __import__("tce_rvos_amd", fromlist=["load_synth_weights"]).load_synth_weights
    load(model, 21)   # leave the shared model as other tests expect it
    model.repack()


def test_graph_cache_is_bounded_lru_over_many_shapes(models):
    """The callers feed whole videos of varying length and caption length: 20 distinct (T, L) shapes must not pile up
    graphs / arenas.  Shapes run eagerly on first sight, are captured when they come back, and the
    cache holds at most `max_graphs` entries."""
    model = models("swin_t_p4w7", 3)
    model.max_graphs, model.graph_after = 3, 1
    H, W = 64, 96
    tgt = [{"size": torch.tensor([H, W])}]
    shapes = [(1 + i % 5, 5 + i // 5) for i in range(20)]
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    outs = {}
    for rnd in range(3):   # round 0 eager, round 1 capture + replay, round 2 replay or re-capture after eviction
        for (T, L) in shapes:
            ids = torch.arange(3, 3 + L)[None]
            o = model([synth_frames(T, H, W, T).cuda()], ids, tgt)["pred_masks"]
            if rnd == 0:
                outs[(T, L)] = o.clone()
            else:
                assert torch.equal(o, outs[(T, L)]), "eager and graph-replay results differ"
            assert len(model._graphs) <= 3
    torch.cuda.synchronize()
    grown = torch.cuda.memory_allocated() - base
    per_graph = max(e[4] for e in model._graphs.values())
    assert grown <= 4 * per_graph + (1 << 30), f"memory grew by {grown / 2**30:.1f} GiB over 60 forwards"
    model.max_graphs, model.graph_after = 6, 1


def test_text_cache_and_video_driver(models):
    """run_video() = the DAVIS clip_size chunking (inference_davis.py:209-256) / whole-video clips
    (inference_ytvos.py:278-295) over the forward + harness kernels, with RoBERTa evaluated once per expression."""
    from tce_rvos_amd.video import run_video
    model = models("swin_t_p4w7", 5)
    N, H, W, H0, W0 = 11, 64, 96, 120, 180
    frames = synth_frames(N, H, W, 17).cuda()
    ids = torch.randint(3, 50000, (1, 9), generator=torch.Generator().manual_seed(2))
    ids[0, 0], ids[0, -1] = 0, 2
    tgt = [{"size": torch.tensor([H, W])}]
    model.text_cache_size = 0
    ref_chunks = []
    for lo in range(0, N, 4):   # the caller's loop written out: one forward per chunk, harness on the oracle side
        o = model([frames[lo:lo + 4]], ids, tgt)
        m, q = O.select_masks(o["pred_logits"].cpu()[0], o["pred_masks"].cpu()[0], (H0, W0))
        ref_chunks.append((m, q))
    model.text_cache_size = 4
    res = run_video(model, frames, ids, (H0, W0), clip_size=4)
    assert len(model._text_cache) == 1
    assert tuple(res["masks"].shape) == (N, H0, W0) and res["masks"].dtype == torch.uint8
    got = res["masks"].cpu().bool()
    want = torch.cat([m for m, _ in ref_chunks], 0)
    assert [int(v) for v in res["best_query"].cpu()] == [int(q) for _, q in ref_chunks]
    assert O.mask_iou(got, want) > 1 - 1e-3
    # cached text features == recomputed ones, bit for bit (same kernels, same inputs)
    a = model([frames[:4]], ids, tgt)["pred_masks"]
    model.text_cache_size = 0
    b = model([frames[:4]], ids, tgt)["pred_masks"]
    assert torch.equal(a, b)
    whole = run_video(model, frames, ids, (H0, W0), clip_size=None)   # YTVOS style: the video is one clip
    assert tuple(whole["masks"].shape) == (N, H0, W0) and whole["best_query"].numel() == 1


def test_run_sharded_world1_on_gpu(models):
    """dist.run_sharded with real clips through the HIP forward (world = 1: the gather is the identity), moving the
    harness's uint8 masks instead of fp32 logits."""
    from tce_rvos_amd import ops
    from tce_rvos_amd.dist import run_sharded
    model = models("swin_t_p4w7", 9)
    H, W = 64, 96
    tgt = [{"size": torch.tensor([H, W])}]
    ids = torch.arange(3, 12)[None]
    clips = [synth_frames(3, H, W, 40 + i).cuda() for i in range(8)]

    def fwd(clip):
        o = model([clip], ids, tgt)
        return ops.select_masks(o["pred_logits"][0], o["pred_masks"][0], (H, W))[0]

    out = run_sharded(fwd, clips)
    assert tuple(out.shape) == (8, 3, H, W) and out.dtype == torch.uint8
    for i in (0, 7):
        assert torch.equal(out[i], fwd(clips[i]))
    empty = run_sharded(fwd, [], like=out[0])
    assert tuple(empty.shape) == (0, 3, H, W)


@pytest.mark.parametrize("group", [1, 2])
def test_two_rank_gloo_rehearsal_on_one_device(group):
    """bench.py's N > 1 control flow (clip sharding, harness kernel, asynchronous all-gather of uint8 masks, barrier,
    max-over-ranks timing) with two processes sharing this box's one GPU: gloo for the collective, TCE_BENCH_ONE_DEVICE=1
    puts both ranks on cuda:0.  Each rank is a fresh child process (nothing here has touched the GPU on its behalf)."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TCE_BENCH_ONE_DEVICE="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                                       "--steps", "4", "--warmup", "2", "--frames", "2", "--height", "96", "--width", "128",
                                       "--tokens", "8", "--no-cpu-baseline", "--no-roofline", "--group", str(group)] +
                                      # group 1 runs WITH the N > 1 variants (the driver's scaling run does): every rank then also times
                                      # 8 clips per forward, gather included, and the line carries value_group8
                                      (["--no-variants"] if group > 1 else []),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    line = json.loads([l for l in outs[0][0].strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["clips_per_step"] == 2 * group and line["value"] > 0
    assert line["collective"]["gathered_shape"][0] == 2 * group and line["collective"]["own_block_matches"]
    assert "all_gather(uint8 masks)" in line["config"]["parallelism"]
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]   # only rank 0 prints the JSON line
    if group == 1:
        assert line["value_group8"] > 0 and line["config"]["clips_per_forward"] == 1   # the default line stays G = 1
    else:
        assert max(line["group_first_clip_vs_b1_max_rel_err"].values()) <= 2e-5


def test_config4_per_rank_shape_through_the_two_rank_flow():
    """BASELINE config 4 = 64 config-2 clips over 8 GPUs = EIGHT clips per rank.  That per-rank shape (Swin-T, T = 5, 360 x 640,
    32 tokens, 8 clips per forward as one launch program) goes through bench.py's N > 1 control flow here with two ranks sharing
    this box's GPU over gloo: 16 clips per step, every rank finds its own block at its own place in the gathered masks (bench.py
    exits non-zero otherwise: the gathered order), and rank 0's first clip of a group equals its own B = 1 forward to 2e-5 of
    the tensor's range.  What stays untested is the hardware only (8 devices, RCCL over xGMI).  Ref: inference_ytvos.py:96-113."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TCE_BENCH_ONE_DEVICE="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                                       "--steps", "3", "--warmup", "2", "--group", "8", "--no-cpu-baseline", "--no-roofline",
                                       "--no-variants"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    line = json.loads([l for l in outs[0][0].strip().splitlines() if l.startswith("{")][-1])
    assert "BASELINE config 2" in line["config"]["workload"] and line["n_gpus"] == 2
    assert line["config"]["clips_per_step"] == 16 and line["config"]["clips_per_forward"] == 8 and line["value"] > 0
    c = line["collective"]
    assert c["gathered_shape"] == [16, 5, 360, 640] and c["own_block_matches"] and c["world"] == 2
    assert c["bytes_per_rank_per_step"] == 8 * 5 * 360 * 640
    rel = line["group_first_clip_vs_b1_max_rel_err"]
    assert max(rel.values()) <= 2e-5, rel


def test_valid_indices_single_frame_path_matches_reference(models):
    """tce_rvos.py:233-243 (A2D / JHMDB: one annotated frame per clip): the backbone sees the clip's 3 frames, everything after
    it frame 1 only (t -> 1).  Against the REFERENCE run with targets[0]['valid_indices'] = 1; eager, captured and replayed."""
    fx = load_npz("e2e_swin_t_valid_idx.npz")
    T, H, W = (int(v) for v in fx["thw"])
    vi = int(fx["valid_index"])
    model = models("swin_t_p4w7", int(fx["weights_salt"]))
    frames = synth_frames(T, H, W, int(fx["frames_seed"])).cuda()
    hid, pooled = torch.from_numpy(fx["text_hidden"])[0].cuda(), torch.from_numpy(fx["text_pooled"])[0].cuda()
    outs = [model.forward_features(frames, hid, pooled, float(H), float(W), select=vi) for _ in range(3)]  # eager, capture, replay
    torch.cuda.synchronize()
    out = outs[0]
    assert tuple(out["pred_masks"].shape) == (1, 1, 5, 18, 25) and tuple(out["memory"].shape) == tuple(fx["out_memory"].shape)
    for k, tol in (("pred_logits", 2e-3), ("pred_boxes", 1e-4), ("reference_points", 1e-4), ("memory", 1e-3)):
        assert (out[k].cpu() - torch.from_numpy(fx["out_" + k])).abs().max().item() < tol, k
    ref = torch.from_numpy(fx["out_pred_masks"])
    d = (out["pred_masks"].cpu() - ref).abs().max().item()
    assert d < 5e-3 and d <= 2e-5 * float(ref.abs().max()), d
    assert O.mask_iou(out["pred_masks"].cpu() > 0, ref > 0) > 1 - 1e-3
    for i in range(3):
        assert (out["aux_outputs"][i]["pred_masks"].cpu() - torch.from_numpy(fx[f"aux{i}_pred_masks"])).abs().max().item() < 5e-3
    for o in outs[1:]:
        for k in ("pred_logits", "pred_boxes", "pred_masks", "memory"):
            assert torch.equal(o[k], out[k]), k
    # through the reference's own boundary: targets[0]['valid_indices'] (token ids; the text encoder differs from the fixture's,
    # so only shapes and the frame count are asserted here)
    ids = torch.randint(3, 50000, (1, 9))
    o2 = model([frames], ids, [{"size": torch.tensor([H, W]), "valid_indices": torch.tensor(vi)}])
    assert tuple(o2["pred_masks"].shape) == (1, 1, 5, 18, 25) and tuple(o2["pred_logits"].shape) == (1, 1, 5, 1)
    with pytest.raises(IndexError):
        model([frames], ids, [{"size": torch.tensor([H, W]), "valid_indices": torch.tensor(T)}])


def test_vis_loss_and_contrastive_outputs_match_reference():
    """A27's optional keys on the HIP path: --vis_loss (visible_embed heads riding in the box MLP's few-row launch ->
    pred_visible, aux levels too) and --contrastive (tce_contrastive_f32) against the reference run with both flags."""
    from tce_rvos_amd import build_model, load_synth_weights
    fx = load_npz("e2e_swin_t_vis_contrastive.npz")
    T, H, W = (int(v) for v in fx["thw"])
    a = _args("swin_t_p4w7")
    a.vis_loss, a.contrastive = True, True
    model, _, _ = build_model(a)
    model = model.cuda().eval()
    load_synth_weights(model, int(fx["weights_salt"]))
    model.repack()
    frames = synth_frames(T, H, W, int(fx["frames_seed"])).cuda()
    hid, pooled = torch.from_numpy(fx["text_hidden"])[0].cuda(), torch.from_numpy(fx["text_pooled"])[0].cuda()
    outs = [model.forward_features(frames, hid, pooled, float(H), float(W)) for _ in range(3)]   # eager, capture, replay
    torch.cuda.synchronize()
    out = outs[0]
    assert set(out) == {"pred_logits", "pred_boxes", "pred_masks", "pred_visible", "contrastive", "aux_outputs", "reference_points", "memory"}
    for k, tol in (("pred_logits", 2e-3), ("pred_boxes", 1e-4), ("pred_visible", 2e-3), ("contrastive", 1e-5), ("pred_masks", 5e-3)):
        ref = torch.from_numpy(fx["out_" + k])
        assert tuple(out[k].shape) == tuple(ref.shape), k
        assert (out[k].cpu() - ref).abs().max().item() < tol, k
    for i in range(3):
        assert (out["aux_outputs"][i]["pred_visible"].cpu() - torch.from_numpy(fx[f"aux{i}_pred_visible"])).abs().max().item() < 2e-3
    for o in outs[1:]:
        for k in ("pred_visible", "contrastive", "pred_masks"):
            assert torch.equal(o[k], out[k]), k
    assert model.hazard_check(frames, torch.randint(3, 50000, (1, 9)).cuda(), (H, W)).clean


def _free_port():
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def test_rccl_world1_bench_path_executes_the_collective():
    """VERDICT r3 #3a: RCCL itself, on device tensors, on the one GPU this box has.  TCE_BENCH_FORCE_DIST=1 makes a world
    of ONE initialise the nccl (= RCCL) process group and send every step's uint8 masks through gather_clip_masks_async ->
    all_gather_into_tensor(async_op=True) -> PendingGather.wait -> barrier -> all_reduce(MAX) -- the N > 1 code path of
    bench.py, not a short-cut.  Fresh child process: nothing has touched the GPU on its behalf before RCCL initialises."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               TCE_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--backend", "nccl", "--steps", "6",
                        "--warmup", "2", "--frames", "2", "--height", "96", "--width", "128", "--tokens", "8",
                        "--no-cpu-baseline", "--no-roofline", "--no-variants"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.strip().splitlines() if l.startswith("{")][-1])
    c = line["collective"]
    assert c["backend"] == "nccl" and c["world"] == 1 and c["forced_world1"] and c["own_block_matches"]
    assert c["gathered_shape"] == [1, 2, 96, 128] and c["bytes_per_rank_per_step"] == 2 * 96 * 128
    assert "RCCL all_gather(uint8 masks)" in line["config"]["parallelism"] and line["value"] > 0


_RCCL_DIST_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ["TCE_DIST_FORCE"] = "1"
from tce_rvos_amd import dist as D
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
g = torch.Generator().manual_seed(0)
local = (torch.rand(3, 5, 24, 40, generator=g) > 0.5).to(torch.uint8).cuda()
out = D.gather_clip_masks(local, 3)                       # blocking form: all_gather_into_tensor on device memory
assert out.data_ptr() != local.data_ptr() and torch.equal(out, local), "blocking gather"
pend = D.gather_clip_masks_async(local, 3)                # async form
assert pend._work is not None, "the collective was short-cut"
assert torch.equal(pend.wait(), local), "async gather"
logits = torch.randn(3, 5, 24, 40, generator=g).cuda()    # fp32 payload through run_sharded (forward = identity)
got = D.run_sharded(lambda c: c, list(logits.unbind(0)))
assert torch.equal(got, logits), "run_sharded"
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert float(t) == 1.5
dist.destroy_process_group()
print("RCCL_WORLD1_OK", torch.cuda.nccl.version())
"""


def test_rccl_world1_dist_module():
    """The sharding module's three entry points (blocking gather, asynchronous gather, run_sharded) + barrier + all_reduce
    through RCCL on device tensors in a world of one (TCE_DIST_FORCE=1), in a fresh child process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _RCCL_DIST_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in p.stdout, (p.stdout[-1000:], p.stderr[-3000:])


# ---------------------------------------------------------------------------------------------------------------
# Round 3: isolation between models, arithmetic stamps, capture budget, clips in flight
# ---------------------------------------------------------------------------------------------------------------
def _small_clip(seed=3, T=3, H=96, W=128, L=9):
    g = torch.Generator().manual_seed(seed)
    frames = synth_frames(T, H, W, seed).cuda()
    hid = torch.randn(L, 768, generator=g).cuda()
    pooled = torch.tanh(torch.randn(768, generator=g)).cuda()
    return frames, hid, pooled, float(H), float(W)


def test_two_models_keep_their_own_packed_streams():
    """packed weight streams used to live in process-global tables that every model's _pack() cleared, so a
    second model's pack freed streams the first model's captured graphs still pointed to.  Now each model owns its routes:
    A's eager and replayed results must not move when B is packed, run, re-packed or freed."""
    from tce_rvos_amd import build_model, load_synth_weights
    a, _, _ = build_model(_args("swin_t_p4w7"))
    b, _, _ = build_model(_args("swin_t_p4w7"))
    a, b = a.cuda().eval(), b.cuda().eval()
    load_synth_weights(a, 1)
    load_synth_weights(b, 2)
    clip = _small_clip()
    # big enough for the rowlin / conv3x3 / fused-FFN routes (>= 12000 stride-4 pixels)
    big = _small_clip(seed=4, T=2, H=320, W=416)
    outs_a = [a.forward_features(*c) for c in (clip, clip, big, big)]   # eager, captured, eager, captured
    ob = b.forward_features(*big)
    b.repack()
    ob2 = b.forward_features(*big)
    ob3 = b.forward_features(*big)
    torch.cuda.synchronize()
    assert torch.equal(ob["pred_masks"], ob2["pred_masks"]) and torch.equal(ob2["pred_masks"], ob3["pred_masks"])
    assert not torch.equal(ob["pred_masks"], outs_a[2]["pred_masks"])
    del b, ob, ob2, ob3
    torch.cuda.empty_cache()
    junk = torch.full((256 << 20,), 7.0, device="cuda")   # whatever B's streams were is overwritten
    again = [a.forward_features(*c) for c in (clip, big)]
    torch.cuda.synchronize()
    del junk
    assert torch.equal(again[0]["pred_masks"], outs_a[0]["pred_masks"])
    assert torch.equal(again[1]["pred_masks"], outs_a[2]["pred_masks"])
    assert torch.equal(outs_a[0]["pred_masks"], outs_a[1]["pred_masks"]) and torch.equal(outs_a[2]["pred_masks"], outs_a[3]["pred_masks"])


def test_gemm_mode_switch_rebuilds_packs_and_graphs(models):
    """packed operands and captured graphs carry the arithmetic they were built in.  Switching the process mode
    (or the model's per-site policy) without repack() must neither replay the old mode's graph nor run fp16-rounded
    streams under the split mode's name."""
    from tce_rvos_amd import ops
    model = models("swin_t_p4w7", 13)
    clip = _small_clip(seed=6, T=2, H=320, W=416)
    x3 = [model.forward_features(*clip)["pred_masks"].clone() for _ in range(3)]
    assert torch.equal(x3[0], x3[2])
    try:
        ops.set_gemm_mode("f16")
        h = [model.forward_features(*clip)["pred_masks"].clone() for _ in range(3)]
    finally:
        ops.set_gemm_mode("f16x3")
    assert torch.equal(h[0], h[2]) and not torch.equal(h[0], x3[0])
    rel = ((h[0] - x3[0]).abs().max() / x3[0].abs().max()).item()
    assert 1e-5 < rel < 5e-2, rel     # fp16-class, not split-class and not garbage
    back = model.forward_features(*clip)["pred_masks"]
    assert torch.equal(back, x3[0])   # lo planes are back: bit-identical to the first split-mode result
    # the per-site policy is part of the stamp too
    model.set_arith_policy({"backbone.mlp": "f16", "encoder.ffn": "f16"})
    p = [model.forward_features(*clip)["pred_masks"].clone() for _ in range(3)]
    model.set_arith_policy({})
    assert torch.equal(p[0], p[2]) and not torch.equal(p[0], x3[0]) and not torch.equal(p[0], h[0])
    assert ((p[0] - x3[0]).abs().max() / x3[0].abs().max()).item() < rel * 1.5
    assert torch.equal(model.forward_features(*clip)["pred_masks"], x3[0])


def test_graph_budget_exhaustion_stays_correct(models, monkeypatch):
    """VERDICT r2 #8b: hipGraph executables are never destroyed, so captures are budgeted per process; once the budget is
    spent new shapes run eagerly for good.  Drive a tiny budget to exhaustion: results stay identical, the state is
    reported, a warning is raised once."""
    import warnings
    from tce_rvos_amd import model as M
    model = models("swin_t_p4w7", 17)
    monkeypatch.setattr(M, "GRAPH_OWN_EXEC", False)   # the legacy path: executables the runtime must never see destroyed
    monkeypatch.setattr(M, "GRAPH_BUDGET", len(M._ALL_GRAPHS) + 2)
    monkeypatch.setattr(M, "_BUDGET_WARNED", False)
    H, W = 64, 96
    tgt = [{"size": torch.tensor([H, W])}]
    ids = torch.arange(3, 10)[None]
    first = {}
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for rnd in range(3):
            for T in (1, 2, 3, 4):
                o = model([synth_frames(T, H, W, T).cuda()], ids, tgt)["pred_masks"]
                if rnd == 0:
                    first[T] = o.clone()
                else:
                    assert torch.equal(o, first[T]), (rnd, T)
    st = model.graph_state()
    assert st["eager_forever"] and st["captured"] == st["budget"]
    assert len(model._graphs) == 2   # two shapes were captured before the budget ran out, two run eagerly
    assert sum("capture budget" in str(w.message) for w in rec) == 1


def test_graph_executables_are_destroyed_on_eviction():
    """600 distinct clip shapes through a 3-entry graph cache in ONE process (a server sweeping shapes): every
    shape's replay equals its eager pass bit for bit, evicted executables are destroyed (owned executables: nodes re-created in a
    fresh graph), the capture budget is never touched and an evicted shape that comes back is captured again.  In a child
    process: a runtime crash fails this test, not the session."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "graph_cycle.py"), "--shapes", "600"], capture_output=True,
                       text=True, timeout=1500)
    assert p.returncode == 0, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    res = json.loads([l for l in p.stdout.strip().splitlines() if l.startswith("{")][-1])
    st = res["graph_state"]
    assert res["own_exec"] and res["max_cached_entries"] <= 3
    assert st["captured"] == 0 and not st["eager_forever"]          # nothing went to the never-destroyed list
    assert st["owned_destroyed"] >= 600 - 3 and st["owned_alive"] <= 3


def test_capture_falls_back_to_eager_when_a_branch_arena_is_too_small(models, monkeypatch):
    """a capture whose branch arena does not fit the shape's kernels must not fail the forward."""
    import warnings
    from tce_rvos_amd import ops
    model = models("swin_t_p4w7", 19)
    clip = _small_clip(seed=8, T=2, H=64, W=96)
    eager = model.forward_features(*clip)["pred_masks"].clone()   # first sighting: eager
    real, calls = ops.Arena, []

    def small(device, nbytes):
        calls.append(nbytes)
        return real(device, (1 << 16) if len(calls) == 2 else nbytes)   # the capture's side arena: far too small

    monkeypatch.setattr(ops, "Arena", small)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        o2 = model.forward_features(*clip)["pred_masks"]   # second sighting: capture attempt -> eager
        o3 = model.forward_features(*clip)["pred_masks"]
    assert any("graph capture" in str(w.message) for w in rec)
    assert torch.equal(o2, eager) and torch.equal(o3, eager)
    assert len(calls) >= 5 and not model._graphs


def test_clips_in_flight_match_one_at_a_time(models):
    """VERDICT r2 #3c: C independent B=1 forwards in flight on C streams (one replay slot each, bench.py
    --clips-in-flight) give, clip for clip, the bits of the one-at-a-time forward."""
    model = models("swin_t_p4w7", 23)
    H, W, T = 96, 128, 3
    tgt = [{"size": torch.tensor([H, W])}]
    ids = torch.arange(3, 12)[None].cuda()
    clips = [synth_frames(T, H, W, 50 + i).cuda() for i in range(4)]
    solo = [model([c], ids, tgt) for c in clips for _ in range(2)][1::2]   # second sighting of each = graph replay, slot 0
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    cur = torch.cuda.current_stream()
    for rnd in range(3):   # round 0: eager per slot, round 1: capture per slot, round 2: two replays in flight
        outs = []
        for base in (0, 2):
            for c in range(2):
                streams[c].wait_stream(cur)
                with torch.cuda.stream(streams[c]):
                    outs.append(model([clips[base + c]], ids, tgt, slot=c))
            for c in range(2):
                cur.wait_stream(streams[c])
        torch.cuda.synchronize()
        for i in range(4):
            for k in ("pred_logits", "pred_boxes", "pred_masks", "memory"):
                assert torch.equal(outs[i][k], solo[i][k]), (rnd, i, k)


@pytest.fixture(scope="module")
def cfg5_oracle(models):
    """BASELINE config 5 (Swin-B, T=10, 480x854) on the synthetic weights, with the oracle's outputs (one CPU forward
    shared by the arithmetic-policy tests)."""
    backbone, T, H, W = "swin_b_p4w7", 10, 480, 854
    torch.set_num_threads(min(32, torch.get_num_threads()))
    model = models(backbone, 11)
    frames = synth_frames(T, H, W, 123)
    g = torch.Generator().manual_seed(7)
    hid = torch.randn(32, 768, generator=g)
    pooled = torch.tanh(torch.randn(768, generator=g))
    sd = {k: v.cpu() for k, v in model.state_dict().items() if not k.startswith("text_encoder.")}
    b = __import__("tce_rvos_amd.config", fromlist=["BACKBONES"]).BACKBONES[backbone]
    cfg = O.OracleConfig(backbone=backbone, embed_dim=b["embed_dim"], depths=b["depths"], num_heads=b["num_heads"])
    with torch.no_grad():
        ref = O.forward(sd, cfg, frames, hid[None], pooled[None], img_size=(H, W))
    return model, (frames.cuda(), hid.cuda(), pooled.cuda(), float(H), float(W)), ref


@pytest.mark.parametrize("policy,rel_bound,iou_bound", [
    # everything downstream of the decoder's queries in single-pass fp16: robustly inside the north star's criterion
    ("cfg5_mixed", 2e-4, 1 - 1e-4),
    # fp16 upstream of the queries too (backbone, encoder): fp16-class error ON the 1e-3 IoU criterion -- 0.99976 on this clip,
    # 0.99846 on bench.py's (profiles/r03_bench_cfg5_swin_b_fast.json): asserted as what it is, not as a pass of the criterion
    ("cfg5_fast", 5e-3, 1 - 3e-3),
    # BASELINE config 5 taken literally, every product one fp16 MFMA: 0.99955 here, 0.99795 on bench.py's clip
    ("all_f16", 1.5e-2, 1 - 3e-3),
])
def test_config5_mixed_fp16_matches_oracle(cfg5_oracle, policy, rel_bound, iou_bound):
    """BASELINE config 5 ("fp16 MFMA") at full size in its reduced-precision arithmetic, per-site maps
    from the committed sensitivity table (profiles/r03_arith_sensitivity_cfg5.txt, tools/arith_sensitivity.py).  Finding: only
    the sites downstream of the decoder's queries take single-pass fp16 with margin; upstream of them the error class
    (2e-3 .. 5e-3 of max|ref|) lands on the IoU criterion and the verdict depends on the clip."""
    model, clip, ref = cfg5_oracle
    assert model._stamp is None or not model.arith_policy
    model.set_arith_policy(policy)
    try:
        outs = [model.forward_features(*clip) for _ in range(3)]   # eager, captured, replayed
        torch.cuda.synchronize()
        assert model.arith_policy and model._stamp[1] == tuple(sorted(model.arith_policy.items()))
        out = outs[-1]
        assert torch.equal(outs[0]["pred_masks"], out["pred_masks"])
        rm = ref["pred_masks"]
        d = (out["pred_masks"].cpu() - rm).abs().max().item()
        scale = rm.abs().max().item()
        iou = O.mask_iou(out["pred_masks"].cpu() > 0, rm > 0)
        dl = (out["pred_logits"].cpu() - ref["pred_logits"]).abs().max().item()
        db = (out["pred_boxes"].cpu() - ref["pred_boxes"]).abs().max().item()
        print(f"config 5, policy {policy}: IoU {iou:.6f}  |d|/max {d / scale:.2e}  logits {dl:.1e}  boxes {db:.1e}")
        assert iou > iou_bound
        assert d <= rel_bound * scale
        assert dl < 5e-3 and db < 2e-3
    finally:
        model.set_arith_policy({})


def test_unit_scale_mask_logits_in_mixed_fp16(models):
    """The O(1)-logit regime (see test_unit_scale_mask_logits_make_the_iou_criterion_bite) in the mixed fp16 policy: more than
    0.1 % of the pixels lie within 1e-2 of the threshold, so the 1e-3 IoU criterion has to be earned by the arithmetic."""
    model = models("swin_t_p4w7", 21)
    with torch.no_grad():
        for k in ("controller.layers.2.weight", "controller.layers.2.bias"):
            model.state_dict(keep_vars=True)[k].mul_(0.2)
    model.repack()
    T, H, W = 5, 180, 320
    frames = synth_frames(T, H, W, 31)
    g = torch.Generator().manual_seed(8)
    hid, pooled = torch.randn(20, 768, generator=g), torch.tanh(torch.randn(768, generator=g))
    sd = {k: v.cpu() for k, v in model.state_dict().items() if not k.startswith("text_encoder.")}
    with torch.no_grad():
        ref = O.forward(sd, O.OracleConfig(), frames, hid[None], pooled[None], img_size=(H, W))
    rm = ref["pred_masks"]
    assert (rm.abs() < 1e-2).float().mean().item() > 1e-3
    try:
        for policy in ("cfg5_mixed", "cfg5_fast", "all_f16"):
            model.set_arith_policy(policy)
            out = model.forward_features(frames.cuda(), hid.cuda(), pooled.cuda(), float(H), float(W))
            torch.cuda.synchronize()
            iou = O.mask_iou(out["pred_masks"].cpu() > 0, rm > 0)
            d = (out["pred_masks"].cpu() - rm).abs().max().item()
            print(f"O(1) logits, policy {policy}: IoU {iou:.6f}  max|d| {d:.2e} (max|ref| {rm.abs().max().item():.2f})")
            if policy == "cfg5_mixed":
                assert iou > 1 - 1e-4          # downstream-of-the-queries sites only: fp32-class masks
            else:
                # fp16 upstream of the queries sits ON the criterion in this regime (cfg5_fast 0.99944, all_f16 0.99899 here)
                assert iou > 1 - 3e-3
    finally:
        model.set_arith_policy({})
        load = __import__("tce_rvos_amd", fromlist=["load_synth_weights"]).load_synth_weights
        load(model, 21)
        model.repack()


@pytest.mark.parametrize("backbone", ["swin_t_p4w7", "resnet50", "video_swin_t_p4w7"])
def test_padded_clip_matches_oracle(models, backbone):
    """VERDICT r2 'missing' #3: a clip zero-padded at the bottom / right (what nested_tensor_from_videos_list produces with
    size_divisibility, util/misc.py:354-377): mask pyramid, masked position maps, valid ratios on every reference point, zero
    value rows at padded positions, key padding masks in the VisionLanguageBlocks' self-attention -- against the oracle run
    with the same pad mask.  Host shape metadata: no device read-back; a foreign NestedTensor costs one and gives the same
    bits; masks that are not a bottom / right border are rejected."""
    from tce_rvos_amd import nested_tensor_from_videos_list, NestedTensor
    model = models(backbone, 31)
    T, hv, wv = 3, 90, 140
    clip = synth_frames(T, hv, wv, 61)
    nt = nested_tensor_from_videos_list([clip.cuda()], size_divisibility=32)
    H, W = nt.tensors.shape[-2:]
    assert (H, W) == (96, 160) and nt.unpadded is False and nt.valid_hw == [(hv, wv)]
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(3, 50000, (1, 10), generator=g)
    ids[0, 0], ids[0, -1] = 0, 2
    tgt = [{"size": torch.tensor([H, W])}]
    outs = [model(nt, ids, tgt) for _ in range(3)]   # eager, captured, replayed
    torch.cuda.synchronize()
    for k in ("pred_logits", "pred_boxes", "pred_masks", "memory"):
        assert torch.equal(outs[0][k], outs[2][k]), k
    hid, pooled = model.forward_text_encoder(ids, "cuda")
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    bb = __import__("tce_rvos_amd.config", fromlist=["BACKBONES"]).BACKBONES[backbone]
    ocfg = O.OracleConfig(backbone=backbone, **{k: bb[k] for k in ("embed_dim", "depths", "num_heads") if k in bb})
    ref = O.forward(sd, ocfg, nt.tensors[0].cpu(), hid.cpu(), pooled.cpu(), img_size=(H, W), pad_mask=nt.mask[0].cpu())
    out = outs[0]
    for k, tol in (("pred_logits", 2e-3), ("pred_boxes", 1e-4), ("reference_points", 1e-4)):
        d = (out[k].cpu() - ref[k]).abs().max().item()
        assert d < tol, (k, d)
    pm, rm = out["pred_masks"].cpu(), ref["pred_masks"]
    d = (pm - rm).abs().max().item()
    print(f"padded clip {hv}x{wv} in {H}x{W}: max|d mask logits| {d:.2e} (max|ref| {rm.abs().max().item():.1f}), "
          f"IoU {O.mask_iou(pm > 0, rm > 0):.6f}")
    # At padded positions the reference's position embedding is sin / cos of -3.1e6 / 10000^(2i/F): chaotic in the last bit of
    # pow (test_pos_sine2d_padded) -- the features of PADDED pixels therefore differ between any two implementations, and
    # the 3x3 convolutions / resamplings of the pixel decoder carry that into a band of valid pixels next to the border.
    # Away from the border (the band is at most one stride-32 cell = 8 mask pixels wide) the masks match like un-padded ones.
    dm = (pm - rm).abs()[0]                      # [T, Q, h4, w4]
    hv4, wv4 = hv // 4, wv // 4
    for margin in (0, 4, 8, 12):
        print(f"  valid region minus {margin:2d} mask pixels at the padded border: max|d| {dm[..., :hv4 - margin, :wv4 - margin].max().item():.2e}")
    scale = rm.abs().max().item()
    assert dm[..., :hv4 - 12, :wv4 - 12].max().item() < 5e-3 + 2e-5 * scale     # measured 2.1e-3 (un-padded small tests: ~3e-3)
    assert d < 2e-3 * scale                                                       # measured 7e-4 of max|ref| in the band
    inner = (slice(None), slice(None), slice(None), slice(0, hv4 - 8), slice(0, wv4 - 8))
    assert O.mask_iou(pm[inner] > 0, rm[inner] > 0) > 1 - 1e-3
    # the un-padded forward of the same frames is a different computation (and the padded one is not a no-op)
    plain = model([nt.tensors[0]], ids, tgt)
    assert not torch.equal(plain["pred_masks"], out["pred_masks"])
    # a NestedTensor without the host metadata: one read-back of the mask, same bits
    foreign = NestedTensor(nt.tensors, nt.mask)
    assert torch.equal(model(foreign, ids, tgt)["pred_masks"], out["pred_masks"])
    bad = nt.mask.clone()
    bad[0, :, 10:20, 10:20] = True
    with pytest.raises(NotImplementedError):
        model(NestedTensor(nt.tensors, bad), ids, tgt)


# ---------------------------------------------------------------------------------------------------------------------
# The launch program itself: race-free by construction (tce_rvos_amd/hazard.py; VERDICT r3 "next round" #1)
# ---------------------------------------------------------------------------------------------------------------------
def _hazard(model, T, H, W, L=32, valid=None, dry=False):
    frames = synth_frames(T, H, W, 11).cuda()
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(3, 50000, (1, L), generator=g).cuda()
    return model.hazard_check(frames, ids, (H, W), valid=valid, dry=dry)


@pytest.mark.parametrize("backbone,T,H,W", [("swin_t_p4w7", 3, 96, 132),       # small: every fallback (un-fused) form
                                            ("swin_t_p4w7", 5, 360, 640),      # BASELINE config 2: the fused forms
                                            ("video_swin_t_p4w7", 8, 384, 640),  # BASELINE config 3
                                            ("resnet50", 1, 360, 640)])        # BASELINE config 1
def test_launch_program_is_race_free(models, backbone, T, H, W):
    """One pass of the clip's launch program on the capture topology (6 streams, 5 arenas) is recorded -- every launch with
    the byte ranges it reads / writes, every fork / join edge -- and any two launches not ordered by happens-before must
    touch disjoint memory.  Needs no timing luck: what a captured graph may run concurrently is exactly what is compared."""
    rep = _hazard(models(backbone, 5), T, H, W)
    print(rep)
    assert rep.launches > 200 and rep.streams >= 4 and rep.unordered_pairs > 1000  # the program really forked
    assert rep.clean, str(rep)


def test_launch_program_race_free_padded_and_long_caption(models):
    m = models("swin_t_p4w7", 5)
    rep = _hazard(m, 3, 96, 160, valid=(90, 140))  # padded clip: valid-region kernels, key-padding masks
    assert rep.clean, str(rep)
    rep = _hazard(m, 2, 64, 96, L=40)  # > 32 tokens: the un-folded cross-attention forms
    assert rep.clean, str(rep)


def test_hazard_checker_sees_a_dropped_join(models, monkeypatch):
    """Negative control on the real program: with every join of the fork / join topology dropped, the checker must report
    conflicts (the text branch's keys, the level slices' consumers, the decoder's outputs ...)."""
    from tce_rvos_amd import pipeline
    m = models("swin_t_p4w7", 5)
    assert _hazard(m, 3, 96, 132).clean  # (warms the per-shape constants up with the real program)
    monkeypatch.setattr(pipeline._Fork, "join", lambda self: None)
    rep = _hazard(m, 3, 96, 132, dry=True)  # recorded only: the broken program is never launched
    torch.cuda.synchronize()
    print(rep)
    assert not rep.clean and rep.n_conflicts > 10


# ---------------------------------------------------------------------------------------------------------------------
# Clip groups: G independent clips as ONE launch program (model.forward_group)
# ---------------------------------------------------------------------------------------------------------------------
def _group_inputs(G, T, H, W, L, seed=70):
    clips = [synth_frames(T, H, W, seed + i).cuda() for i in range(G)]
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, 50000, (G, L), generator=g).cuda()  # G different captions of one length
    return clips, ids


GROUP_KEYS = ("pred_logits", "pred_boxes", "pred_masks", "memory", "reference_points")


@pytest.mark.parametrize("backbone,G,T,H,W,L", [("swin_t_p4w7", 2, 3, 96, 132, 9),      # small: the un-fused forms
                                                ("swin_t_p4w7", 3, 2, 64, 96, 40),      # > 32 tokens: un-folded text cross-attention
                                                ("swin_t_p4w7", 2, 5, 360, 640, 32),    # BASELINE config 2 shapes
                                                ("swin_t_p4w7", 4, 5, 360, 640, 32),
                                                ("swin_t_p4w7", 8, 5, 360, 640, 32),    # BASELINE config 4's per-GPU batch; 256 caption tokens
                                                ("video_swin_t_p4w7", 2, 4, 96, 128, 9),  # 3-D windows: one launch per clip
                                                ("resnet50", 3, 1, 96, 128, 9),
                                                ("resnet50", 8, 1, 96, 128, 20),        # 160 caption tokens: tiled-GEMM text layers
                                                ("swin_t_p4w7", 5, 2, 64, 96, 33)])
def test_clip_group_matches_one_clip_at_a_time(models, backbone, G, T, H, W, L):
    """A group of G clips is block-diagonal across clips at every stage that looks across frames or at the caption, so each
    clip's outputs are its B = 1 forward's: same arithmetic per row, but some sites take another kernel route at the group's
    row count (a different summation order) -- hence a tolerance, 2e-5 of the tensor's range, not bit equality.  Eager pass,
    capture and replay of the group must agree bit for bit with each other."""
    model = models(backbone, 31)
    clips, ids = _group_inputs(G, T, H, W, L)
    tgt = [{"size": torch.tensor([H, W])}]
    solo = [model([clips[g]], ids[g:g + 1], tgt) for g in range(G)]
    solo = [{k: v.clone() for k, v in o.items() if k in GROUP_KEYS} for o in solo]
    runs = []
    for _ in range(3):  # eager, capture, replay
        outs = model.forward_group(clips, ids, tgt)
        torch.cuda.synchronize()
        runs.append([{k: o[k].clone() for k in GROUP_KEYS if k in o} for o in outs])
    assert len(runs[0]) == G
    for g in range(G):
        for k in solo[g]:
            ref, got = solo[g][k], runs[0][g][k]
            assert got.shape == ref.shape, (g, k, got.shape, ref.shape)
            tol = 2e-5 * float(ref.abs().max()) + 1e-6
            err = float((got - ref).abs().max())
            assert err <= tol, (g, k, err, tol)
            for r in runs[1:]:
                assert torch.equal(r[g][k], got), (g, k, "replay != eager")
        # mask agreement over the pixels whose sign means something: a logit inside the tolerance band above may fall on either
        # side of zero in either run (on this 96 x 128 clip a mask has ~120 pixels: one such flip would read as IoU 0.992)
        pm = solo[g]["pred_masks"]
        sure = pm.abs() > 2e-5 * float(pm.abs().max()) + 1e-6
        m_ref, m_got = (pm > 0) & sure, (runs[0][g]["pred_masks"] > 0) & sure
        inter, union = (m_ref & m_got).sum().item(), (m_ref | m_got).sum().item()
        assert union == 0 or inter / union > 0.9999


@pytest.mark.parametrize("backbone,G,T,H,W", [("swin_t_p4w7", 3, 3, 96, 132), ("swin_t_p4w7", 4, 5, 360, 640),
                                              ("video_swin_t_p4w7", 2, 4, 96, 128), ("resnet50", 4, 1, 96, 128)])
def test_expressions_of_one_clip_share_the_backbone(models, backbone, G, T, H, W):
    """forward_group with the SAME clip tensor G times (G expressions of one video): the backbone runs once, its maps are
    repeated, the rest is the group program.  Each result = the B = 1 forward of (clip, that caption); the program is race-free."""
    model = models(backbone, 31)
    clips, ids = _group_inputs(1, T, H, W, 9)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, 50000, (G, 9), generator=g).cuda()
    tgt = [{"size": torch.tensor([H, W])}]
    solo = [{k: v.clone() for k, v in model([clips[0]], ids[i:i + 1], tgt).items() if k in GROUP_KEYS} for i in range(G)]
    runs = []
    for _ in range(3):  # eager, capture, replay
        outs = model.forward_group([clips[0]] * G, ids, tgt)
        torch.cuda.synchronize()
        runs.append([{k: o[k].clone() for k in GROUP_KEYS} for o in outs])
    for i in range(G):
        for k in GROUP_KEYS:
            ref, got = solo[i][k], runs[0][i][k]
            assert got.shape == ref.shape
            assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6, (i, k)
            assert torch.equal(runs[1][i][k], got) and torch.equal(runs[2][i][k], got), (i, k, "replay != eager")
    rep = model.hazard_check(clips[0], ids, (H, W), groups=G, shared=True)
    assert rep.clean, str(rep)


def test_run_video_expressions_matches_run_video(models):
    """The drivers' loop over a video's expressions (inference_ytvos.py:96-113): grouped by caption length, shared backbone;
    every expression's masks / chosen query = run_video's for that expression alone."""
    from tce_rvos_amd.video import run_video, run_video_expressions
    model = models("swin_t_p4w7", 31)
    H, W, H0, W0 = 96, 128, 180, 240
    frames = synth_frames(7, H, W, 90).cuda()
    caps = ["the left zebra", "a person walking a dog", "the right zebra", "the small dog", "a car"]   # lengths 5, 7, 5, 5, 4
    res = run_video_expressions(model, frames, caps, (H0, W0), clip_size=4, max_group=2)
    assert len(res) == len(caps)
    for c, r in zip(caps, res):
        ref = run_video(model, frames, c, (H0, W0), clip_size=4)
        assert torch.equal(r["best_query"], ref["best_query"]), c
        assert r["masks"].shape == ref["masks"].shape == (7, H0, W0)
        assert (r["masks"] != ref["masks"]).float().mean().item() < 1e-4, c
        assert float((r["pred_logits"] - ref["pred_logits"]).abs().max()) < 1e-4


def test_clip_group_clips_do_not_see_each_other(models):
    """Changing clip 1 (frames and caption) must leave clip 0's outputs bit-identical: nothing crosses the clips of a group."""
    model = models("swin_t_p4w7", 31)
    H, W = 96, 132
    tgt = [{"size": torch.tensor([H, W])}]
    clips, ids = _group_inputs(2, 3, H, W, 9)
    a = model.forward_group(clips, ids, tgt)[0]
    a = {k: a[k].clone() for k in GROUP_KEYS}
    other, ids2 = _group_inputs(2, 3, H, W, 9, seed=123)
    ids_b = torch.cat([ids[:1], ids2[1:]], 0)
    b = model.forward_group([clips[0], other[1]], ids_b, tgt)[0]
    for k in GROUP_KEYS:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("backbone,G,T,H,W", [("swin_t_p4w7", 2, 3, 96, 132), ("swin_t_p4w7", 4, 5, 360, 640),
                                              ("video_swin_t_p4w7", 2, 8, 384, 640)])
def test_clip_group_launch_program_is_race_free(models, backbone, G, T, H, W):
    model = models(backbone, 5)
    clips, ids = _group_inputs(G, T, H, W, 32)
    rep = model.hazard_check(torch.cat(clips, 0), ids, (H, W), groups=G)
    print(rep)
    assert rep.launches > 200 and rep.unordered_pairs > 1000
    assert rep.clean, str(rep)


def test_clip_group_rejects_ragged_groups(models):
    model = models("swin_t_p4w7", 31)
    tgt = [{"size": torch.tensor([96, 128])}]
    clips, ids = _group_inputs(2, 3, 96, 128, 9)
    with pytest.raises(ValueError):
        model.forward_group([clips[0], clips[1][:2]], ids, tgt)
    with pytest.raises(ValueError):
        model.forward_group(clips, ids[:1], tgt)
    with pytest.raises(ValueError):   # captions of unequal token length go in separate groups
        model.forward_group(clips, ["a dog", "a dog running on the grass"], tgt)
    # string captions of one length: tokenised like forward's
    outs = model.forward_group(clips, ["the left zebra", "the left zebra"], tgt)
    solo = model([clips[1]], ["the left zebra"], tgt)
    assert float((outs[1]["pred_masks"] - solo["pred_masks"]).abs().max()) <= 2e-5 * float(solo["pred_masks"].abs().max()) + 1e-6
