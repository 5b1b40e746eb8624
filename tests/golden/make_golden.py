"""Generates tests/golden/*.npz by RUNNING THE REFERENCE in the build container (CPU).

Run:  cd /root/repo && python tests/golden/make_golden.py
Needs /root/reference (read-only) -- never runs on the GPU box.  The files it writes are data
(inputs, weights of deliberately tiny model configurations, and the reference's outputs).

Fixtures
  msda_cases.npz       reference MSDA core (ms_deform_attn_func.py:67-87) on the reference's own
                       test case (models/ops/test.py:21-26, seed 3) + cases with out-of-range
                       sampling locations and a realistic slice; forward outputs and the gradients
                       wrt value / locations / weights for a seeded grad_output
  interp_cases.npz     F.interpolate nearest / bilinear(align_corners=False) at the odd size pairs
                       the model hits (12x20->23x40, 23x40->11x20 ...)
  e2e_swin_t_small.npz full ReferFormer.forward of the reference, real Swin-T architecture, T=3, 72x100
                       (non-multiples of 4*7, odd merges), flags of scripts/dist_test_davis.sh; weights are
                       tce_rvos_amd.weights.synth_state_dict (a function of key+shape+salt, not stored);
                       frames = randn(seed); text encoder outputs stored (third-party RoBERTa, random init)
  e2e_vswin_t_small.npz same with Video-Swin-T, T=9 (> window depth 8: temporal shift path)
  e2e_swin_t_cfg2.npz  BASELINE config 2 at full size (T=5, 360x640): outputs only
  e2e_vswin_t_cfg3.npz BASELINE config 3 at full size (Video-Swin-T, T=8, 384x640): outputs only
  e2e_swin_b_cfg5.npz  BASELINE config 5 at full size (Swin-B, T=10, 480x854): outputs only (+ statedict_swin_b.json)
  e2e_swin_t_padded.npz a PADDED clip: T=3 frames of 90x140 through the reference's own nested_tensor_from_videos_list(...,
                       size_divisibility=32) -> 96x160 with its pad mask (util/misc.py:354-377), i.e. the mask pyramid, the
                       masked cumulative position maps, valid ratios, zero-filled MSDA values and key-padding masks of the
                       reference itself; outputs + memory + the stride-8 position map of frame 0 + mask features
  statedict_*.json     the reference's state-dict keys/shapes (the drop-in checkpoint contract); *_plain = without
                       --with_box_refine/--f_token/--qtrans
  harness_cases.npz    caller harness H (inference_ytvos.py:238-250): logits+masks -> thresholded mask
  e2e_swin_t_valid_idx.npz  the A2D / JHMDB single-frame path: T=3 frames of 72x100 with targets[0]['valid_indices'] = 1
                       (tce_rvos.py:233-243: everything after the backbone sees that one frame; t -> 1)
  e2e_swin_t_vis_contrastive.npz  the optional output keys of --vis_loss / --contrastive (pred_visible, contrastive) +
                       statedict_swin_t_vis.json (visible_embed.* in the checkpoint contract)
  perop_swin_t.npz, perop_vswin_t.npz, perop_noqtrans.npz  (round 5, SURVEY 8c list (ii)-(vii)) inputs and outputs of the
                       reference's OWN sub-modules, captured by forward hooks while the reference runs a small clip with the
                       synthetic weights (weights regenerate from the salt): shifted + padded SwinTransformerBlock and its
                       WindowAttention (with the -100 mask), PatchMerging at an odd size, VisionLanguageFusionModule,
                       the four VisionLanguageBlocks (sr 8 / 4 / 2 / 1), FrameTokenLayer, an encoder layer, decoder layers with
                       2-d and 4-d reference points (qtrans on; perop_noqtrans: off), dynamic_mask_with_coords;
                       WindowAttention3D at T = 9 (8-frame windows, shifted, masked) and at T = 3 ((3,7,7) windows, [:N,:N] table)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy()


def gen_msda():
    rh.import_reference()
    from models.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    out = {}
    # case 0: the reference's own test (models/ops/test.py:21-26,47-52), float path
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = [(6, 4), (3, 2)]
    S = sum(h * w for h, w in shapes)
    torch.manual_seed(3)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    w = torch.rand(N, Lq, M, L, P) + 1e-5
    w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
    cases = [(shapes, value, loc, w)]
    # case 1: out-of-range locations, 2 frames
    torch.manual_seed(11)
    N, M, D, Lq, L, P = 2, 2, 32, 7, 2, 4
    shapes = [(6, 4), (3, 2)]
    S = sum(h * w for h, w in shapes)
    value = torch.randn(N, S, M, D)
    loc = torch.rand(N, Lq, M, L, P, 2) * 1.6 - 0.3
    w = torch.softmax(torch.randn(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    cases.append((shapes, value, loc, w))
    # case 2: realistic slice (config-2 level shapes), 64 queries, out-of-range included
    torch.manual_seed(12)
    N, M, D, Lq, L, P = 1, 8, 32, 64, 4, 4
    shapes = [(23, 40), (12, 20), (6, 10), (3, 5)]
    S = sum(h * w for h, w in shapes)
    value = torch.randn(N, S, M, D)
    loc = torch.rand(N, Lq, M, L, P, 2) * 1.2 - 0.1
    w = torch.softmax(torch.randn(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    cases.append((shapes, value, loc, w))
    for i, (shapes, value, loc, w) in enumerate(cases):
        o = ms_deform_attn_core_pytorch(value, shapes, loc, w)
        out[f"c{i}_shapes"] = np.asarray(shapes, dtype=np.int64)
        out[f"c{i}_value"] = _np(value)
        out[f"c{i}_loc"] = _np(loc)
        out[f"c{i}_w"] = _np(w)
        out[f"c{i}_out"] = _np(o)
        # gradients of the reference's own core (autograd through its grid_sample formulation) for a seeded grad_output:
        # the pin of the backward op (the reference's test checks its CUDA backward against exactly this, test.py:60-86)
        torch.manual_seed(100 + i)
        go = torch.randn_like(o)
        v_, l_, w_ = (t.clone().requires_grad_(True) for t in (value, loc, w))
        gv, gl, gw = torch.autograd.grad(ms_deform_attn_core_pytorch(v_, shapes, l_, w_), (v_, l_, w_), go)
        out[f"c{i}_gout"], out[f"c{i}_gvalue"], out[f"c{i}_gloc"], out[f"c{i}_gw"] = _np(go), _np(gv), _np(gl), _np(gw)
    out["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(HERE, "msda_cases.npz"), **out)


def gen_interp():
    torch.manual_seed(5)
    out = {}
    pairs = [((12, 20), (23, 40)), ((23, 40), (45, 80)), ((45, 80), (90, 160)), ((23, 40), (11, 20)),
             ((45, 80), (11, 20)), ((90, 160), (11, 20)), ((3, 4), (5, 7)), ((9, 13), (4, 6)), ((18, 25), (2, 3))]
    for i, (a, b) in enumerate(pairs):
        x = torch.randn(2, 3, *a)
        out[f"p{i}_in"] = _np(x)
        out[f"p{i}_size"] = np.asarray(b)
        out[f"p{i}_nearest"] = _np(F.interpolate(x, size=b, mode="nearest"))
        out[f"p{i}_bilinear"] = _np(F.interpolate(x, size=b, mode="bilinear", align_corners=False))
    out["n_pairs"] = np.asarray(len(pairs))
    np.savez_compressed(os.path.join(HERE, "interp_cases.npz"), **out)


def _synth_inputs(T, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(T, 3, H, W, generator=g)


def gen_e2e(name, backbone, T, H, W, seed, store_stages=True, stage_keys=None, pad_div=None):
    """Real architecture (Swin-T / Video-Swin-T, hidden 256, 4+4 layers), weights from
    tce_rvos_amd.weights.synth_state_dict (keyed by state-dict name, so they need not be stored).
    pad_div: the [T,3,H,W] clip goes through the reference's nested_tensor_from_videos_list(size_divisibility=pad_div)
    and the model receives the resulting NestedTensor (padded frames + pad mask)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tce_rvos_amd.weights import load_synth_weights
    args = rh.reference_args(backbone)
    model = rh.build_reference_model(args, seed=0, roberta_layers=1)
    load_synth_weights(model, salt=seed)
    cap = {}
    model.text_encoder.register_forward_hook(
        lambda m, i, o: cap.update(hid=o.last_hidden_state.detach(), pool=o.pooler_output.detach()))
    stages = {}
    model.transformer.encoder.register_forward_hook(lambda m, i, o: stages.update(memory=o.detach()))
    model.pixel_decoder.register_forward_hook(lambda m, i, o: stages.update(mask_features=o.detach()))
    model.backbone[0].register_forward_hook(
        lambda m, i, o: stages.update({f"backbone{k}": v.tensors.detach() for k, v in o.items()}))
    frames = _synth_inputs(T, H, W, seed + 1)
    extra = {}
    if pad_div is None:
        samples, size = [frames], (H, W)
    else:
        from util.misc import nested_tensor_from_videos_list  # the reference's own (util/misc.py:354-377)
        samples = nested_tensor_from_videos_list([frames], size_divisibility=pad_div)
        size = tuple(int(v) for v in samples.tensors.shape[-2:])
        extra = {"padded_hw": np.asarray(size), "pad_mask": _np(samples.mask[0])}
        model.backbone[1].register_forward_hook(
            lambda m, i, o: stages.setdefault("pos_calls", []).append(o.detach()))  # PositionEmbeddingSine2D per level
    with torch.no_grad():
        out = model(samples, ["synthetic"], [{"size": torch.tensor(size)}])
    pos_calls = stages.pop("pos_calls", None)
    if pos_calls is not None:  # call order = sorted backbone levels (stride 4, 8, 16, 32), then the extra level
        stages["pos1_frame0"] = pos_calls[1][0]
        stages["mask_features"] = stages["mask_features"][:1]  # frame 0 only (fixture size)
    fx = {"text_hidden": _np(cap["hid"]), "text_pooled": _np(cap["pool"]),
          "thw": np.asarray([T, H, W]), "frames_seed": np.asarray(seed + 1), "weights_salt": np.asarray(seed)}
    fx.update(extra)
    for k in ("pred_logits", "pred_boxes", "pred_masks", "reference_points"):
        fx["out_" + k] = _np(out[k])
    fx["out_memory_sum"] = np.asarray(out["memory"].double().sum().item())
    fx["out_memory_abs_sum"] = np.asarray(out["memory"].double().abs().sum().item())
    for i, a in enumerate(out["aux_outputs"]):
        for k, v in a.items():
            if k != "pred_masks" or store_stages:
                fx[f"aux{i}_{k}"] = _np(v)
    if store_stages:
        if stage_keys is None or "memory" in stage_keys:
            fx["out_memory"] = _np(out["memory"])
        for k, v in stages.items():
            if stage_keys is None or k in stage_keys:
                fx["stage_" + k] = _np(v)
    fx["cfg_backbone"] = np.asarray(backbone)
    np.savez_compressed(os.path.join(HERE, name), **fx)
    print(name, "pred_masks", tuple(out["pred_masks"].shape), os.path.getsize(os.path.join(HERE, name)))
    return model


def gen_statedict_manifest(model, name):
    import json
    man = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()
           if not k.startswith("text_encoder.")}
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)


def gen_plain_manifest():
    """State-dict contract of the reference WITHOUT --with_box_refine / --f_token / --qtrans (shared heads are listed
    under every level's name, no decoder.bbox_embed alias, no frame-token parameters)."""
    import opts
    args = opts.get_args_parser().parse_args(["--binary", "--freeze_text_encoder", "--backbone", "swin_t_p4w7"])
    args.masks, args.device = True, "cpu"
    model = rh.build_reference_model(args, roberta_layers=1)
    gen_statedict_manifest(model, "statedict_swin_t_plain.json")


def gen_harness():
    """inference_ytvos.py:238-250 restated with the same PyTorch primitives the caller uses."""
    torch.manual_seed(21)
    out = {}
    for i, (t, q, h, w, H0, W0) in enumerate([(3, 5, 18, 25, 70, 99), (5, 5, 23, 40, 90, 160)]):
        logits = torch.randn(1, t, q, 1)
        masks = torch.randn(1, t, q, h, w) * 3
        pred_logits = logits[0]
        pred_masks = masks[0]
        pred_scores = pred_logits.sigmoid().mean(0)
        max_scores, _ = pred_scores.max(-1)
        _, max_ind = max_scores.max(-1)
        max_inds = max_ind.repeat(t)
        pm = pred_masks[range(t), max_inds, ...].unsqueeze(0)
        pm = F.interpolate(pm, size=(H0, W0), mode="bilinear", align_corners=False)
        pm = (pm.sigmoid() > 0.5).squeeze(0)
        out[f"h{i}_logits"] = _np(logits)
        out[f"h{i}_masks"] = _np(masks)
        out[f"h{i}_size"] = np.asarray([H0, W0])
        out[f"h{i}_best"] = np.asarray(int(max_ind))
        out[f"h{i}_out"] = _np(pm)
    out["n_cases"] = np.asarray(2)
    np.savez_compressed(os.path.join(HERE, "harness_cases.npz"), **out)


def gen_resnet():
    """BASELINE config 1 (ResNet-50, T=1): a small clip with stages + the full 360x640 frame (outputs only)."""
    m = gen_e2e("e2e_resnet50_small.npz", "resnet50", T=2, H=72, W=100, seed=6,
                stage_keys=("backbone0", "backbone3", "memory"))
    gen_statedict_manifest(m, "statedict_resnet50.json")
    gen_e2e("e2e_resnet50_cfg1.npz", "resnet50", T=1, H=360, W=640, seed=8, store_stages=False)


def gen_round4():
    """Round 4: what was added or enlarged since round 1, pinned by the reference itself."""
    gen_e2e("e2e_swin_t_padded.npz", "swin_t_p4w7", T=3, H=90, W=140, seed=14, pad_div=32,
            stage_keys=("memory", "mask_features", "pos1_frame0"))
    gen_e2e("e2e_vswin_t_cfg3.npz", "video_swin_t_p4w7", T=8, H=384, W=640, seed=10, store_stages=False)
    m = gen_e2e("e2e_swin_b_cfg5.npz", "swin_b_p4w7", T=10, H=480, W=854, seed=12, store_stages=False)
    gen_statedict_manifest(m, "statedict_swin_b.json")


def _arr(v):
    """A hook argument as an array (None and non-numeric values are skipped by the caller)."""
    if torch.is_tensor(v):
        return _np(v.to(torch.float32) if v.dtype == torch.bool else v)
    if isinstance(v, (int, float, bool)):
        return np.asarray(v)
    if isinstance(v, (list, tuple)) and v and all(isinstance(x, (int, float)) for x in v):
        return np.asarray(v)
    return None


def _hook_modules(model, names, fx, first_only=True):
    """Forward hooks on the named sub-modules of the REFERENCE model: positional inputs -> `<tag>_in<i>`, keyword inputs ->
    `<tag>_kw_<name>`, the output -> `<tag>_out` (tag = the module path with dots as underscores), first call only."""
    mods = dict(model.named_modules())
    handles = []
    for name in names:
        tag = name.replace(".", "_")

        def hook(m, args, kwargs, out, tag=tag):
            if first_only and tag + "_out" in fx:
                return
            for i, a in enumerate(args):
                v = _arr(a)
                if v is not None:
                    fx[f"{tag}_in{i}"] = v
            for k, a in kwargs.items():
                v = _arr(a)
                if v is not None:
                    fx[f"{tag}_kw_{k}"] = v
            o = out[0] if isinstance(out, (tuple, list)) else out
            fx[tag + "_out"] = _np(o)
            for extra in ("H", "W"):  # SwinTransformerBlock carries its grid as attributes set by BasicLayer
                if hasattr(m, extra) and isinstance(getattr(m, extra), int):
                    fx[f"{tag}_attr_{extra}"] = np.asarray(getattr(m, extra))

        handles.append(mods[name].register_forward_hook(hook, with_kwargs=True))
    return handles


def gen_perop(name, backbone, T, H, W, seed, modules, extra_args=(), wrap_mask_head=False):
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tce_rvos_amd.weights import load_synth_weights
    args = rh.reference_args(backbone)
    if extra_args == "noqtrans":
        args.qtrans = False
    model = rh.build_reference_model(args, seed=0, roberta_layers=1)
    load_synth_weights(model, salt=seed)
    fx = {"thw": np.asarray([T, H, W]), "frames_seed": np.asarray(seed + 1), "weights_salt": np.asarray(seed),
          "cfg_backbone": np.asarray(backbone)}
    _hook_modules(model, modules, fx)
    model.text_encoder.register_forward_hook(  # third-party RoBERTa (random init): its outputs are an INPUT of the path
        lambda m, i, o: fx.update(text_hidden=_np(o.last_hidden_state), text_pooled=_np(o.pooler_output)))
    if wrap_mask_head:
        orig = model.dynamic_mask_with_coords

        def wrapped(mask_features, mask_head_params, reference_points, targets):
            o = orig(mask_features, mask_head_params, reference_points, targets)
            if "maskhead_out" not in fx:
                fx.update(maskhead_in0=_np(mask_features), maskhead_in1=_np(mask_head_params), maskhead_in2=_np(reference_points),
                          maskhead_size=_np(targets[0]["size"]), maskhead_out=_np(o))
            return o

        model.dynamic_mask_with_coords = wrapped
    frames = _synth_inputs(T, H, W, seed + 1)
    with torch.no_grad():
        out = model([frames], ["synthetic"], [{"size": torch.tensor((H, W))}])
    fx["out_pred_masks"] = _np(out["pred_masks"])
    missing = [m for m in modules if m.replace(".", "_") + "_out" not in fx]
    assert not missing, missing
    np.savez_compressed(os.path.join(HERE, name), **fx)
    print(name, len(fx), "arrays", os.path.getsize(os.path.join(HERE, name)))


def gen_valid_idx():
    """tce_rvos.py:233-243: one annotated frame per clip."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tce_rvos_amd.weights import load_synth_weights
    T, H, W, seed, vi = 3, 72, 100, 16, 1
    model = rh.build_reference_model(rh.reference_args("swin_t_p4w7"), seed=0, roberta_layers=1)
    load_synth_weights(model, salt=seed)
    cap = {}
    model.text_encoder.register_forward_hook(
        lambda m, i, o: cap.update(hid=o.last_hidden_state.detach(), pool=o.pooler_output.detach()))
    frames = _synth_inputs(T, H, W, seed + 1)
    with torch.no_grad():
        out = model([frames], ["synthetic"], [{"size": torch.tensor((H, W)), "valid_indices": torch.tensor(vi)}])
    fx = {"text_hidden": _np(cap["hid"]), "text_pooled": _np(cap["pool"]), "thw": np.asarray([T, H, W]),
          "frames_seed": np.asarray(seed + 1), "weights_salt": np.asarray(seed), "valid_index": np.asarray(vi),
          "cfg_backbone": np.asarray("swin_t_p4w7")}
    for k in ("pred_logits", "pred_boxes", "pred_masks", "reference_points", "memory"):
        fx["out_" + k] = _np(out[k])
    for i, a in enumerate(out["aux_outputs"]):
        for k, v in a.items():
            fx[f"aux{i}_{k}"] = _np(v)
    np.savez_compressed(os.path.join(HERE, "e2e_swin_t_valid_idx.npz"), **fx)
    print("e2e_swin_t_valid_idx.npz pred_masks", tuple(out["pred_masks"].shape))


def gen_vis_contrastive():
    """--vis_loss --contrastive (tce_rvos.py:62-63,318-319,336-365,512-521): the optional output keys pred_visible / contrastive."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tce_rvos_amd.weights import load_synth_weights
    T, H, W, seed = 3, 72, 100, 22
    model = rh.build_reference_model(rh.reference_args("swin_t_p4w7", extra=("--vis_loss", "--contrastive")), seed=0, roberta_layers=1)
    load_synth_weights(model, salt=seed)
    gen_statedict_manifest(model, "statedict_swin_t_vis.json")
    cap = {}
    model.text_encoder.register_forward_hook(
        lambda m, i, o: cap.update(hid=o.last_hidden_state.detach(), pool=o.pooler_output.detach()))
    frames = _synth_inputs(T, H, W, seed + 1)
    with torch.no_grad():
        out = model([frames], ["synthetic"], [{"size": torch.tensor((H, W))}])
    fx = {"text_hidden": _np(cap["hid"]), "text_pooled": _np(cap["pool"]), "thw": np.asarray([T, H, W]),
          "frames_seed": np.asarray(seed + 1), "weights_salt": np.asarray(seed), "cfg_backbone": np.asarray("swin_t_p4w7")}
    for k in ("pred_logits", "pred_boxes", "pred_masks", "pred_visible", "contrastive", "reference_points"):
        fx["out_" + k] = _np(out[k])
    for i, a in enumerate(out["aux_outputs"]):
        fx[f"aux{i}_pred_visible"] = _np(a["pred_visible"])
    np.savez_compressed(os.path.join(HERE, "e2e_swin_t_vis_contrastive.npz"), **fx)
    print("e2e_swin_t_vis_contrastive.npz", {k: tuple(out[k].shape) for k in ("pred_visible", "contrastive")}, sorted(out))


def gen_round5():
    gen_valid_idx()
    gen_vis_contrastive()
    b = "backbone.0.body."
    gen_perop("perop_swin_t.npz", "swin_t_p4w7", T=3, H=72, W=100, seed=18, wrap_mask_head=True, modules=[
        b + "layers.0.blocks.1", b + "layers.0.blocks.1.attn", b + "layers.1.blocks.0", b + "layers.1.downsample",
        "fusion_module", "pixel_decoder.cross_attn_1", "pixel_decoder.cross_attn_2", "pixel_decoder.cross_attn_3",
        "pixel_decoder.cross_attn_4", "transformer.encoder.layers.0.ftoken_layers", "transformer.encoder.layers.0",
        "transformer.decoder.layers.0", "transformer.decoder.layers.1"])
    gen_perop("perop_noqtrans.npz", "swin_t_p4w7", T=3, H=72, W=100, seed=18, extra_args="noqtrans",
              modules=["transformer.decoder.layers.0", "transformer.decoder.layers.1"])
    gen_perop("perop_vswin_t.npz", "video_swin_t_p4w7", T=9, H=72, W=100, seed=20,
              modules=[b + "layers.0.blocks.1.attn"])
    gen_perop("perop_vswin_t_short.npz", "video_swin_t_p4w7", T=3, H=72, W=100, seed=20,
              modules=[b + "layers.0.blocks.1.attn"])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "round5":
        torch.set_num_threads(8)
        gen_round5()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round4":
        torch.set_num_threads(8)
        gen_round4()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "resnet":
        torch.set_num_threads(8)
        gen_resnet()
        sys.exit(0)
    torch.set_num_threads(8)
    gen_msda()
    gen_interp()
    gen_harness()
    m = gen_e2e("e2e_swin_t_small.npz", "swin_t_p4w7", T=3, H=72, W=100, seed=0)
    gen_statedict_manifest(m, "statedict_swin_t.json")
    m = gen_e2e("e2e_vswin_t_small.npz", "video_swin_t_p4w7", T=9, H=72, W=100, seed=2,
                stage_keys=("backbone1", "backbone3", "memory"))
    gen_statedict_manifest(m, "statedict_vswin_t.json")
    # BASELINE config 2 at full size: only the outputs are kept (inputs/weights regenerate from seeds)
    gen_e2e("e2e_swin_t_cfg2.npz", "swin_t_p4w7", T=5, H=360, W=640, seed=4, store_stages=False)
    gen_plain_manifest()
    gen_resnet()
    gen_round4()
    gen_round5()
    print("done")
