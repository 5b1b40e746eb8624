"""Clip front-end (Resize(360) + ToTensor + Normalize): host tables on the CPU, kernels on the GPU, both against
Pillow itself (oracle/frontend_ref.py) -- integer arithmetic, so the bar is bit-exact."""
import numpy as np
import pytest
import torch

from oracle import frontend_ref as FR

SIZES = [(720, 1280), (480, 854), (240, 320), (360, 640), (1080, 607), (37, 53), (361, 359)]


def _frames(T, H, W, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(T, H, W, 3), dtype=np.uint8)
    base[0, : H // 2] = 255  # saturated / flat regions exercise the clipping and the rounding
    base[-1, :, : W // 3] = 0
    return base


def _apply_tables(frame, size=360):
    """numpy application of the product's host tables: validates bilinear_coeffs without a GPU."""
    from tce_rvos_amd.frontend import PRECISION_BITS, bilinear_coeffs, resize_output_size
    H, W, _ = frame.shape
    h, w = resize_output_size(H, W, size)
    ch, bh, _ = bilinear_coeffs(W, w)
    cv, bv, _ = bilinear_coeffs(H, h)
    half = 1 << (PRECISION_BITS - 1)
    tmp = np.empty((H, w, 3), dtype=np.uint8)
    x = frame.astype(np.int64)
    for xo in range(w):
        x0, n = bh[xo]
        acc = (x[:, x0:x0 + n, :] * ch[xo, :n].astype(np.int64)[None, :, None]).sum(1) + half
        tmp[:, xo, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
    out = np.empty((h, w, 3), dtype=np.uint8)
    t = tmp.astype(np.int64)
    for yo in range(h):
        y0, n = bv[yo]
        acc = (t[y0:y0 + n] * cv[yo, :n].astype(np.int64)[:, None, None]).sum(0) + half
        out[yo] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


@pytest.mark.parametrize("H,W", SIZES)
def test_host_tables_reproduce_pillow(H, W):
    from tce_rvos_amd.frontend import resize_output_size
    assert resize_output_size(H, W) == FR.resize_size(H, W)
    f = _frames(1, H, W, H * 7 + W)[0]
    assert np.array_equal(_apply_tables(f), FR.resized_u8(f))


def test_normalise_lut_matches_totensor_normalize():
    from tce_rvos_amd.frontend import normalise_lut
    lut = normalise_lut()
    f = np.tile(np.arange(256, dtype=np.uint8), 2).reshape(1, 512, 1).repeat(3, 2).repeat(360, 0)  # [360,512,3]: no resize
    ref = FR.transform(np.ascontiguousarray(f))
    for c in range(3):
        assert torch.equal(ref[c, 0, :256], lut[c])


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", SIZES)
def test_frontend_kernels_bit_exact_vs_pillow(H, W):
    from tce_rvos_amd.frontend import ClipFrontEnd
    T = 3
    frames = _frames(T, H, W, H + W)
    fe = ClipFrontEnd(360)
    out = fe(torch.from_numpy(frames).cuda())
    torch.cuda.synchronize()
    ref = torch.stack([FR.transform(frames[t]) for t in range(T)], 0)
    assert out.shape == ref.shape
    assert torch.equal(out.cpu(), ref)


@pytest.mark.gpu
def test_frontend_feeds_the_model_boundary():
    """decoded frames -> front-end -> model -> harness, all on the GPU, equals the same chain fed by the Pillow path."""
    import argparse
    from tce_rvos_amd import build_model, ops
    from tce_rvos_amd.frontend import ClipFrontEnd
    frames = _frames(2, 144, 200, 5)
    model, _, _ = build_model(argparse.Namespace(backbone="swin_t_p4w7", with_box_refine=True, binary=True, f_token=8,
                                                 qtrans=True, text_encoder_layers=1))
    model = model.cuda().eval()
    ids = torch.tensor([[0, 11, 12, 13, 2]])
    a = ClipFrontEnd(72)(torch.from_numpy(frames).cuda())
    b = torch.stack([FR.transform(frames[t], 72) for t in range(2)], 0).cuda()
    assert torch.equal(a, b)
    tgt = [{"size": torch.tensor(list(a.shape[-2:]))}]
    oa, ob = model([a], ids, tgt), model([b], ids, tgt)
    ma, _ = ops.select_masks(oa["pred_logits"][0], oa["pred_masks"][0], (144, 200))
    mb, _ = ops.select_masks(ob["pred_logits"][0], ob["pred_masks"][0], (144, 200))
    assert torch.equal(ma, mb) and ma.shape == (2, 144, 200)
