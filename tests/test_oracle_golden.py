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


def _run_e2e(fixture, manifest_name, backbone):
    fx = load_npz(fixture)
    T, H, W = (int(v) for v in fx["thw"])
    sd = synth_sd_from_manifest(manifest_name, int(fx["weights_salt"]))
    frames = synth_frames(T, H, W, int(fx["frames_seed"]))
    cfg = O.OracleConfig(backbone=backbone)
    with torch.no_grad():
        out = O.forward(sd, cfg, frames, torch.from_numpy(fx["text_hidden"]), torch.from_numpy(fx["text_pooled"]),
                        img_size=(H, W), return_stages=True)
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
