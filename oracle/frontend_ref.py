"""TEST INFRASTRUCTURE (CPU oracle) -- the reference callers' per-frame transform (inference_ytvos.py:38-42):
    T.Compose([T.Resize(360), T.ToTensor(), T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])

torchvision is absent from this image; for a PIL input its three transforms are, as published:
  Resize(int)  -> PIL.Image.resize((w', h'), BILINEAR), shorter edge -> size, other edge int(size * long / short)
  ToTensor     -> uint8 HWC -> CHW float32 / 255
  Normalize    -> (x - mean[c]) / std[c] in float32
The resampling arithmetic itself is NOT restated here: the real dependency, Pillow (installed in this image, same on
the GPU box), is called, so the product kernels are pinned against Pillow's own output bit for bit.
Only tests/ may import this module."""
import numpy as np
import torch
from PIL import Image


def resize_size(h, w, size=360):
    short, long_ = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long_ / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def transform(frame_u8: np.ndarray, size=360) -> torch.Tensor:
    """frame_u8 [H, W, 3] uint8 RGB -> [3, h, w] float32 (one frame, as the caller's loop does)."""
    img = Image.fromarray(frame_u8, mode="RGB")
    h, w = resize_size(img.size[1], img.size[0], size)
    if (h, w) != (img.size[1], img.size[0]):
        img = img.resize((w, h), Image.BILINEAR)
    x = torch.from_numpy(np.array(img, dtype=np.uint8, copy=True)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean = torch.as_tensor([0.485, 0.456, 0.406], dtype=torch.float32).view(3, 1, 1)
    std = torch.as_tensor([0.229, 0.224, 0.225], dtype=torch.float32).view(3, 1, 1)
    return x.sub_(mean).div_(std)


def resized_u8(frame_u8: np.ndarray, size=360) -> np.ndarray:
    img = Image.fromarray(frame_u8, mode="RGB")
    h, w = resize_size(img.size[1], img.size[0], size)
    return np.array(img.resize((w, h), Image.BILINEAR) if (h, w) != (img.size[1], img.size[0]) else img)
