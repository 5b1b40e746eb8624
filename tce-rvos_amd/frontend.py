"""Clip front-end on the GPU: decoded uint8 RGB frames -> the model's normalised fp32 clip.

Mirrors the reference callers' transform (inference_ytvos.py:38-42, inference_davis.py:39-43):
    T.Compose([T.Resize(360), T.ToTensor(), T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
applied per frame to `Image.open(path).convert('RGB')` and stacked (inference_ytvos.py:279-287).  JPEG decoding
stays with the caller; this module takes the decoded frames as one uint8 tensor [T, H, W, 3].

torchvision (absent here) resizes a PIL image with Pillow's Image.resize(BILINEAR); Pillow's algorithm (Resample.c:
precompute_coeffs / normalize_coeffs_8bpc / ImagingResampleHorizontal_8bpc / Vertical_8bpc) is restated here: the
coefficient tables on the host in float64, the two integer passes in csrc/frontend.hip.  Results are bit-identical
to Pillow + torch (tests/test_frontend_*.py check against PIL itself).
"""
import math

import numpy as np
import torch

from ._lib import check, lib
from .ops import _stream

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def resize_output_size(h, w, size=360):
    """torchvision.transforms.Resize(int): the shorter edge becomes `size`, the other int(size * long / short)
    (no max_size); unchanged when the shorter edge already equals `size`."""
    short, long_ = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long_ / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def bilinear_coeffs(in_size, out_size):
    """Pillow precompute_coeffs (triangle filter, support 1) + normalize_coeffs_8bpc.
    Returns (coef int32 [out, ksize], bounds int32 [out, 2] = (first input index, tap count), ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    coef = np.zeros((out_size, ksize), dtype=np.float64)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.arange(xmax, dtype=np.float64)
        wgt = np.abs((x + xmin - center + 0.5) * ss)
        wgt = np.where(wgt < 1.0, 1.0 - wgt, 0.0)
        ww = 0.0
        for v in wgt:  # Pillow accumulates left to right in double
            ww += float(v)
        if ww != 0.0:
            wgt = wgt / ww
        coef[xx, :xmax] = wgt
        bounds[xx] = (xmin, xmax)
    fixed = coef * float(1 << PRECISION_BITS)
    icoef = np.where(fixed < 0, np.trunc(-0.5 + fixed), np.trunc(0.5 + fixed)).astype(np.int32)
    return icoef, bounds, ksize


def normalise_lut():
    """[3, 256] fp32: ToTensor (uint8 -> float / 255) then Normalize ((x - mean) / std), in torch's own fp32 ops."""
    v = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)
    mean = torch.tensor(MEAN, dtype=torch.float32).view(3, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(3, 1)
    return ((v[None, :] - mean) / std).contiguous()


class ClipFrontEnd:
    """frames_u8 [T, H, W, 3] (CUDA uint8, RGB) -> [T, 3, h, w] fp32 normalised, (h, w) = Resize(size) of (H, W)."""

    def __init__(self, size=360):
        self.size = size
        self._tables = {}
        self._lut = None

    def _table(self, n_in, n_out, device):
        key = (n_in, n_out, str(device))
        t = self._tables.get(key)
        if t is None:
            c, b, k = bilinear_coeffs(n_in, n_out)
            t = (torch.from_numpy(c).to(device), torch.from_numpy(b).to(device), k)
            self._tables[key] = t
        return t

    def __call__(self, frames_u8):
        if frames_u8.dtype != torch.uint8 or not frames_u8.is_cuda:
            raise TypeError(f"ClipFrontEnd: expected a CUDA uint8 tensor, got {frames_u8.dtype} on {frames_u8.device}")
        if frames_u8.dim() != 4 or frames_u8.shape[-1] != 3 or not frames_u8.is_contiguous():
            raise ValueError("ClipFrontEnd: frames must be contiguous uint8 [T, H, W, 3]")
        T, H, W, _ = frames_u8.shape
        dev = frames_u8.device
        h, w = resize_output_size(H, W, self.size)
        if self._lut is None or self._lut.device != dev:
            self._lut = normalise_lut().to(dev)
        ch, bh, kh = self._table(W, w, dev)
        cv, bv, kv = self._table(H, h, dev)
        tmp = torch.empty(T, H, w, 3, dtype=torch.uint8, device=dev)
        out = torch.empty(T, 3, h, w, dtype=torch.float32, device=dev)
        check(lib().tce_resize_h_u8(frames_u8.data_ptr(), ch.data_ptr(), bh.data_ptr(), tmp.data_ptr(), T * H, W, w, kh,
                                    _stream()), "tce_resize_h_u8")
        check(lib().tce_resize_v_norm_f32(tmp.data_ptr(), cv.data_ptr(), bv.data_ptr(), self._lut.data_ptr(),
                                          out.data_ptr(), T, H, w, h, kv, _stream()), "tce_resize_v_norm_f32")
        return out
