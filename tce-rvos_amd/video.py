"""Whole-video drivers around the per-clip forward.

Two loops of the reference's callers, with the model call and the caller harness H on the GPU kernels:

* `run_video(..., clip_size=32)`  -- inference_davis.py:209-256: the video is cut into consecutive chunks of `clip_size`
  frames (the last one shorter), each chunk is one B=1 forward, the best query is chosen PER CHUNK from that chunk's
  mean class score, its masks are resized to the original size, and the per-chunk results are concatenated.
* `run_video(..., clip_size=None)` -- inference_ytvos.py:278-321 (the non-`keep_fps` branch): the whole video is one
  clip, whatever its length.

* `run_video_expressions(...)` -- the loop AROUND those two in the reference's drivers (inference_ytvos.py:96-113,
  inference_davis.py:150-208: every expression of a video is run over the same frames): expressions whose captions tokenise to
  one length go through `model.forward_group` with the SAME clip tensor, so the backbone runs once per chunk and the rest of the
  program is shared by the group; each expression's result is what `run_video` returns for it.

A video's chunks share the caption, so with `model.text_cache_size > 0` RoBERTa runs once per expression instead of
once per chunk (the reference recomputes it inside every forward).  Chunks of one length share a captured hipGraph
(model._graphs is an LRU over shapes); a last, shorter chunk runs eagerly unless its shape comes back.
"""
from typing import Optional

import torch

from . import ops


@torch.no_grad()
def run_video(model, frames: torch.Tensor, caption, origin_hw, clip_size: Optional[int] = 32, threshold: float = 0.5):
    """frames [N,3,H,W] float32 on the GPU (already resized + normalised: frontend.py); caption: str or LongTensor
    [1,L] of token ids; origin_hw = (H0, W0) of the decoded video.
    Returns dict(masks uint8 [N,H0,W0] (1 = object), best_query int32 [n_chunks], pred_logits [N,K], pred_boxes [N,4])."""
    if frames.dim() != 4 or frames.shape[1] != 3:
        raise ValueError("run_video: frames must be [N,3,H,W]")
    n = frames.shape[0]
    if n == 0:
        raise ValueError("run_video: empty video")
    H, W = int(frames.shape[-2]), int(frames.shape[-1])
    target = [{"size": torch.tensor([H, W])}]
    cap = [caption] if isinstance(caption, str) else caption
    step = n if not clip_size else int(clip_size)
    masks, best, logits, boxes = [], [], [], []
    for lo in range(0, n, step):
        clip = frames[lo:lo + step]
        out = model([clip], cap, target)
        pl, pm = out["pred_logits"][0], out["pred_masks"][0]
        m, b = ops.select_masks(pl, pm, origin_hw, threshold)  # harness H: query choice + resize + sigmoid + threshold
        masks.append(m)
        best.append(b)
        idx = b.long().expand(pl.shape[0])
        ar = torch.arange(pl.shape[0], device=pl.device)
        logits.append(pl[ar, idx])
        boxes.append(out["pred_boxes"][0][ar, idx])
    ops.check_range(frames.device)  # the driver synchronises here anyway: surface a tripped range flag now
    return {"masks": torch.cat(masks, 0), "best_query": torch.cat(best, 0), "pred_logits": torch.cat(logits, 0),
            "pred_boxes": torch.cat(boxes, 0)}


def _collect(out, origin_hw, threshold, acc):
    pl, pm = out["pred_logits"][0], out["pred_masks"][0]
    m, b = ops.select_masks(pl, pm, origin_hw, threshold)  # harness H: query choice + resize + sigmoid + threshold
    idx = b.long().expand(pl.shape[0])
    ar = torch.arange(pl.shape[0], device=pl.device)
    acc["masks"].append(m)
    acc["best_query"].append(b)
    acc["pred_logits"].append(pl[ar, idx])
    acc["pred_boxes"].append(out["pred_boxes"][0][ar, idx])


@torch.no_grad()
def run_video_expressions(model, frames: torch.Tensor, captions, origin_hw, clip_size: Optional[int] = 32,
                          threshold: float = 0.5, max_group: int = 4):
    """Every expression of ONE video.  frames as in run_video; captions: list of str (or of LongTensor [1,L]).  Returns a list with
    one run_video-style dict per caption, in order.  Captions of equal token length are grouped (at most `max_group` per forward)."""
    if frames.dim() != 4 or frames.shape[1] != 3 or frames.shape[0] == 0:
        raise ValueError("run_video_expressions: frames must be a non-empty [N,3,H,W]")
    n = frames.shape[0]
    H, W = int(frames.shape[-2]), int(frames.shape[-1])
    target = [{"size": torch.tensor([H, W])}]
    ids = [c if torch.is_tensor(c) else model._tokenise([c], frames.device)[0] for c in captions]
    buckets = {}
    for i, t in enumerate(ids):
        buckets.setdefault(int(t.shape[1]), []).append(i)
    step = n if not clip_size else int(clip_size)
    accs = [{"masks": [], "best_query": [], "pred_logits": [], "pred_boxes": []} for _ in captions]
    for members in buckets.values():
        for g0 in range(0, len(members), max(1, int(max_group))):
            grp = members[g0:g0 + max(1, int(max_group))]
            tok = torch.cat([ids[i].to(frames.device) for i in grp], 0)
            for lo in range(0, n, step):
                clip = frames[lo:lo + step]
                outs = model.forward_group([clip] * len(grp), tok, target)  # one tensor, len(grp) captions: shared backbone
                for i, out in zip(grp, outs):
                    _collect(out, origin_hw, threshold, accs[i])
    ops.check_range(frames.device)
    return [{k: torch.cat(v, 0) for k, v in a.items()} for a in accs]
