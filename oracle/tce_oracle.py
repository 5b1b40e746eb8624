"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU (torch fp32) restatement of the TCE-RVOS per-clip forward pass, written from
the reference's semantics (file:line citations point into /root/reference).
It is a flat functional program over a state dict that uses the reference's
parameter names, so the same weights can be fed to the reference (in the
survey container), to this oracle, and to the HIP product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module, and there only as the checker / the timed CPU baseline.

Parity status: PINNED.  tests/golden/*.npz hold outputs of the reference itself
(generated in the build container by tests/golden/make_golden.py); the
`-m "not gpu"` tests check this oracle against them (per-op and end-to-end).
The text encoder (HF RoBERTa, third party, absent from /root/reference) is NOT
restated: the oracle takes `last_hidden_state` / `pooler_output` as inputs
("parity unpinned" for the text encoder only).

B = 1 clip per forward (what every inference caller of the reference does).
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    # backbone (swin_transformer.py:687-745 / video_swin_transformer.py:733-779)
    backbone: str = "swin_t_p4w7"
    embed_dim: int = 96
    depths: Tuple[int, ...] = (2, 2, 6, 2)
    num_heads: Tuple[int, ...] = (3, 6, 12, 24)
    window_size: int = 7
    video_window: Tuple[int, int, int] = (8, 7, 7)
    # transformer (opts.py:40-60)
    hidden_dim: int = 256
    nheads: int = 8
    num_feature_levels: int = 4
    enc_layers: int = 4
    dec_layers: int = 4
    dim_feedforward: int = 2048
    enc_n_points: int = 4
    dec_n_points: int = 4
    num_queries: int = 5
    f_token: int = 8
    qtrans: bool = True
    with_box_refine: bool = True
    # mask head (opts.py:68-74)
    mask_dim: int = 256
    controller_layers: int = 3
    dynamic_mask_channels: int = 8
    rel_coord: bool = True
    vlblock: bool = True
    aux_loss: bool = True
    num_classes: int = 1
    vis_loss: bool = False     # --vis_loss: visible_embed heads (tce_rvos.py:62-63,336-363)
    contrastive: bool = False  # --contrastive: contrastive_cal (tce_rvos.py:318-319,512-521)

    @property
    def is_video_swin(self):
        return "video_swin" in self.backbone

    @property
    def is_resnet(self):
        return self.backbone.startswith("resnet")

    @property
    def num_channels(self):
        if self.is_resnet:  # backbone.py:68
            return [256, 512, 1024, 2048]
        return [self.embed_dim * 2 ** i for i in range(len(self.depths))]


# --------------------------------------------------------------------------------------
# primitive ops (PyTorch documented semantics)
# --------------------------------------------------------------------------------------
def linear(x, w, b=None):
    return F.linear(x, w, b)


def layer_norm(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def nearest_index(out_size: int, in_size: int) -> Tensor:
    """PyTorch legacy 'nearest': src = min(floor(dst * (in/out)), in-1), scale computed in fp32."""
    scale = torch.tensor(float(in_size) / float(out_size), dtype=torch.float32)
    dst = torch.arange(out_size, dtype=torch.float32)
    return torch.clamp(torch.floor(dst * scale).long(), max=in_size - 1)


def interp_nearest(x: Tensor, size: Tuple[int, int]) -> Tensor:
    """x [..., H, W] -> [..., size]; F.interpolate(mode='nearest') semantics."""
    iy = nearest_index(size[0], x.shape[-2])
    ix = nearest_index(size[1], x.shape[-1])
    return x[..., iy, :][..., :, ix]


def _bilinear_taps(out_size: int, in_size: int):
    """align_corners=False source taps: src = max((dst+0.5)*scale-0.5, 0)."""
    scale = torch.tensor(float(in_size) / float(out_size), dtype=torch.float32)
    dst = torch.arange(out_size, dtype=torch.float32)
    src = torch.clamp((dst + 0.5) * scale - 0.5, min=0.0)
    i0 = torch.floor(src).long()
    i0 = torch.clamp(i0, max=in_size - 1)
    i1 = torch.clamp(i0 + 1, max=in_size - 1)
    lam = src - i0.to(torch.float32)
    return i0, i1, lam


def interp_bilinear(x: Tensor, size: Tuple[int, int]) -> Tensor:
    """x [..., H, W] -> [..., size]; F.interpolate(mode='bilinear', align_corners=False)."""
    y0, y1, ly = _bilinear_taps(size[0], x.shape[-2])
    x0, x1, lx = _bilinear_taps(size[1], x.shape[-1])
    ly = ly[:, None]
    top = x[..., y0, :]
    bot = x[..., y1, :]
    rows = top * (1.0 - ly) + bot * ly          # [..., oh, W]
    left = rows[..., :, x0]
    right = rows[..., :, x1]
    return left * (1.0 - lx) + right * lx


def mha(query: Tensor, key: Tensor, value: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor,
        nheads: int, key_padding_mask: Optional[Tensor] = None) -> Tensor:
    """nn.MultiheadAttention forward, seq-first (L, N, E), packed in_proj [3E, E] in q,k,v order,
    q scaled by head_dim**-0.5 before QK^T, key_padding_mask [N, S] (True = ignore) -> -inf."""
    L, N, E = query.shape
    S = key.shape[0]
    d = E // nheads
    q = linear(query, in_w[:E], in_b[:E])
    k = linear(key, in_w[E:2 * E], in_b[E:2 * E])
    v = linear(value, in_w[2 * E:], in_b[2 * E:])
    q = q.reshape(L, N * nheads, d).transpose(0, 1) * (float(d) ** -0.5)
    k = k.reshape(S, N * nheads, d).transpose(0, 1)
    v = v.reshape(S, N * nheads, d).transpose(0, 1)
    attn = torch.bmm(q, k.transpose(1, 2))  # [N*h, L, S]
    if key_padding_mask is not None and bool(key_padding_mask.any()):
        m = key_padding_mask.view(N, 1, 1, S).expand(N, nheads, 1, S).reshape(N * nheads, 1, S)
        attn = attn.masked_fill(m, float("-inf"))
    attn = torch.softmax(attn, dim=-1)
    out = torch.bmm(attn, v)  # [N*h, L, d]
    out = out.transpose(0, 1).reshape(L, N, E)
    return linear(out, out_w, out_b)


def inverse_sigmoid(x: Tensor, eps: float = 1e-5) -> Tensor:
    """util/misc.py:555-559"""
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


# --------------------------------------------------------------------------------------
# position encodings (models/position_encoding.py)
# --------------------------------------------------------------------------------------
def pos_sine_2d(mask: Tensor, num_pos_feats: int = 128, temperature: float = 10000.0) -> Tensor:
    """position_encoding.py:64-84 (normalize=True, scale=2*pi, the -0.5 shift).  mask [N,H,W] bool -> [N,2F,H,W]."""
    not_mask = ~mask
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    eps = 1e-6
    scale = 2 * math.pi
    y_embed = (y_embed - 0.5) / (y_embed[:, -1:, :] + eps) * scale
    x_embed = (x_embed - 0.5) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def pos_sine_1d(mask: Tensor, num_pos_feats: int = 256, temperature: float = 10000.0) -> Tensor:
    """position_encoding.py:28-45 (normalize=True, no -0.5).  mask [B,L] -> [B,C,L]."""
    not_mask = ~mask
    x_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = x_embed / (x_embed[:, -1:] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / num_pos_feats)
    pos_x = x_embed[:, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, 0::2].sin(), pos_x[:, :, 1::2].cos()), dim=3).flatten(2)
    return pos_x.permute(0, 2, 1)


# --------------------------------------------------------------------------------------
# multi-scale deformable attention core (models/ops/src/cuda/ms_deform_im2col_cuda.cuh:34-85,421-452)
# --------------------------------------------------------------------------------------
def msda_core(value: Tensor, shapes: List[Tuple[int, int]], loc: Tensor, weights: Tensor) -> Tensor:
    """value [N,S,M,D], loc [N,Lq,M,L,P,2] (x,y in [0,1] units), weights [N,Lq,M,L,P] -> [N,Lq,M*D].

    The native rule: h_im = y*H - 0.5, w_im = x*W - 0.5; a sample contributes iff
    -1 < h_im < H and -1 < w_im < W; each of the 4 corners is dropped individually when outside."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = torch.zeros(N, Lq, M, D, dtype=value.dtype)
    start = 0
    n_idx = torch.arange(N).view(N, 1, 1, 1)
    m_idx = torch.arange(M).view(1, 1, M, 1)
    for lvl, (H, W) in enumerate(shapes):
        v = value[:, start:start + H * W]  # [N, HW, M, D]
        x = loc[:, :, :, lvl, :, 0] * W - 0.5  # [N,Lq,M,P]
        y = loc[:, :, :, lvl, :, 1] * H - 0.5
        ok = (y > -1) & (x > -1) & (y < H) & (x < W)
        y0 = torch.floor(y)
        x0 = torch.floor(x)
        ly, lx = y - y0, x - x0
        hy, hx = 1 - ly, 1 - lx
        y0, x0 = y0.long(), x0.long()
        acc = torch.zeros(N, Lq, M, P, D, dtype=value.dtype)
        for dy, dx, wgt in ((0, 0, hy * hx), (0, 1, hy * lx), (1, 0, ly * hx), (1, 1, ly * lx)):
            yy, xx = y0 + dy, x0 + dx
            inb = ok & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1))  # [N,Lq,M,P]
            g = v[n_idx, idx, m_idx]  # [N,Lq,M,P,D]
            acc = acc + g * (wgt * inb.to(value.dtype)).unsqueeze(-1)
        out = out + (acc * weights[:, :, :, lvl, :].unsqueeze(-1)).sum(3)
        start += H * W
    return out.reshape(N, Lq, M * D)


def msda_core_backward(value: Tensor, shapes: List[Tuple[int, int]], loc: Tensor, weights: Tensor,
                       grad_out: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """Gradients of msda_core wrt (value, loc, weights) for a given grad_out [N,Lq,M*D]: what the native backward returns
    (ms_deform_attn_cuda.cu:105-186; per-corner rule ms_deform_im2col_cuda.cuh:87-160).  msda_core is built from
    differentiable torch ops whose derivative IS that rule (floor and the in-range masks carry no gradient; d/dx of the
    bilinear weights gives hh (v2 - v1) + lh (v4 - v3), scaled by W through x = loc_x * W - 0.5), so autograd states it."""
    with torch.enable_grad():
        v = value.detach().clone().requires_grad_(True)
        p = loc.detach().clone().requires_grad_(True)
        w = weights.detach().clone().requires_grad_(True)
        out = msda_core(v, shapes, p, w)
        gv, gp, gw = torch.autograd.grad(out, (v, p, w), grad_out.reshape(out.shape))
    return gv, gp, gw


def msda_module(sd: Dict[str, Tensor], pre: str, query: Tensor, ref: Tensor, src: Tensor,
                shapes: List[Tuple[int, int]], padding_mask: Optional[Tensor], M: int, L: int, P: int):
    """MSDeformAttn.forward, ops/modules/ms_deform_attn.py:79-117.  Returns (out, sampling_locations, weights)."""
    N, Lq, C = query.shape
    S = src.shape[1]
    value = linear(src, sd[pre + "value_proj.weight"], sd[pre + "value_proj.bias"])
    if padding_mask is not None:
        value = value.masked_fill(padding_mask[..., None], 0.0)
    value = value.view(N, S, M, C // M)
    off = linear(query, sd[pre + "sampling_offsets.weight"], sd[pre + "sampling_offsets.bias"]).view(N, Lq, M, L, P, 2)
    aw = linear(query, sd[pre + "attention_weights.weight"], sd[pre + "attention_weights.bias"]).view(N, Lq, M, L * P)
    aw = torch.softmax(aw, -1).view(N, Lq, M, L, P)
    if ref.shape[-1] == 2:
        normalizer = torch.tensor([[w, h] for (h, w) in shapes], dtype=torch.float32)
        loc = ref[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    out = msda_core(value, shapes, loc, aw)
    out = linear(out, sd[pre + "output_proj.weight"], sd[pre + "output_proj.bias"])
    return out, loc, aw


# --------------------------------------------------------------------------------------
# Swin backbone (models/swin_transformer.py)
# --------------------------------------------------------------------------------------
def rel_pos_index(ws: int) -> Tensor:
    """swin_transformer.py:107-117 ('ij' meshgrid)."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij"))
    cf = coords.flatten(1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_attn_mask(Hp: int, Wp: int, ws: int, shift: int) -> Tensor:
    """swin_transformer.py:370-388: region ids on the PADDED grid, additive -100 where ids differ."""
    img = torch.zeros(Hp, Wp)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(am != 0, torch.tensor(-100.0), torch.tensor(0.0))


def window_attention(sd, pre, xw: Tensor, nh: int, ws: int, mask: Optional[Tensor]) -> Tensor:
    """WindowAttention.forward swin_transformer.py:127-158.  xw [nWB, N, C]."""
    B_, N, C = xw.shape
    d = C // nh
    qkv = linear(xw, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(B_, N, 3, nh, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (float(d) ** -0.5), qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    table = sd[pre + "relative_position_bias_table"]
    bias = table[rel_pos_index(ws).view(-1)].view(N, N, nh).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, nh, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, nh, N, N)
    attn = torch.softmax(attn, dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return linear(x, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def swin_block(sd, pre, x: Tensor, H: int, W: int, nh: int, ws: int, shift: int, attn_mask: Tensor) -> Tensor:
    """SwinTransformerBlock.forward swin_transformer.py:202-258.  x [B, H*W, C]."""
    B, Ltok, C = x.shape
    shortcut = x
    x = layer_norm(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"]).view(B, H, W, C)
    pad_r = (ws - W % ws) % ws
    pad_b = (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))  # zeros AFTER the norm, not masked
    Hp, Wp = H + pad_b, W + pad_r
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    aw = window_attention(sd, pre + "attn.", xw, nh, ws, attn_mask if shift > 0 else None)
    x = aw.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = x[:, :H, :W, :].reshape(B, H * W, C)
    x = shortcut + x
    y = layer_norm(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    y = linear(y, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])
    y = F.gelu(y)  # exact erf form
    y = linear(y, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    return x + y


def patch_merging(sd, pre, x: Tensor, H: int, W: int) -> Tensor:
    """PatchMerging.forward swin_transformer.py:273-299."""
    B, Ltok, C = x.shape
    x = x.view(B, H, W, C)
    if H % 2 == 1 or W % 2 == 1:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.view(B, -1, 4 * C)
    x = layer_norm(x, sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    return linear(x, sd[pre + "reduction.weight"])


def swin_backbone(sd, cfg: OracleConfig, frames: Tensor, pre="backbone.0.body.") -> List[Tensor]:
    """SwinTransformer.forward swin_transformer.py:595-617.  frames [B,3,H,W] -> 4 maps [B,Ci,hi,wi]."""
    ps = 4
    _, _, H0, W0 = frames.shape
    x = frames
    if W0 % ps:
        x = F.pad(x, (0, ps - W0 % ps))
    if H0 % ps:
        x = F.pad(x, (0, 0, 0, ps - H0 % ps))
    x = F.conv2d(x, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"], stride=ps)
    B, C, Wh, Ww = x.shape
    x = x.flatten(2).transpose(1, 2)
    x = layer_norm(x, sd[pre + "patch_embed.norm.weight"], sd[pre + "patch_embed.norm.bias"])
    outs = []
    ws = cfg.window_size
    H, W = Wh, Ww
    for i, depth in enumerate(cfg.depths):
        Hp = int(math.ceil(H / ws)) * ws
        Wp = int(math.ceil(W / ws)) * ws
        am = shift_attn_mask(Hp, Wp, ws, ws // 2)
        for j in range(depth):
            x = swin_block(sd, f"{pre}layers.{i}.blocks.{j}.", x, H, W, cfg.num_heads[i], ws,
                           0 if j % 2 == 0 else ws // 2, am)
        xo = layer_norm(x, sd[f"{pre}norm{i}.weight"], sd[f"{pre}norm{i}.bias"])
        outs.append(xo.view(B, H, W, -1).permute(0, 3, 1, 2).contiguous())
        if i < len(cfg.depths) - 1:
            x = patch_merging(sd, f"{pre}layers.{i}.downsample.", x, H, W)
            H, W = (H + 1) // 2, (W + 1) // 2
    return outs


# --------------------------------------------------------------------------------------
# ResNet-50 backbone (models/backbone.py; body = torchvision resnet50, absent from /root/reference and from this image:
# restated from its published definition -- He et al. 2016 bottleneck, "v1.5" stride on the 3x3 -- and anchored on the
# reference's call site backbone.py:92-96 and its own FrozenBatchNorm2d :20-56, IntermediateLayerGetter names :64-74)
# --------------------------------------------------------------------------------------
def frozen_bn(sd, pre, x: Tensor) -> Tensor:
    """backbone.py:46-56: scale = w * rsqrt(var + 1e-5); bias = b - mean * scale; x * scale + bias."""
    scale = sd[pre + "weight"] * (sd[pre + "running_var"] + 1e-5).rsqrt()
    bias = sd[pre + "bias"] - sd[pre + "running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def resnet_bottleneck(sd, pre, x: Tensor, stride: int) -> Tensor:
    idt = x
    if pre + "downsample.0.weight" in sd:
        idt = frozen_bn(sd, pre + "downsample.1.", F.conv2d(x, sd[pre + "downsample.0.weight"], stride=stride))
    y = F.relu(frozen_bn(sd, pre + "bn1.", F.conv2d(x, sd[pre + "conv1.weight"])))
    y = F.relu(frozen_bn(sd, pre + "bn2.", F.conv2d(y, sd[pre + "conv2.weight"], stride=stride, padding=1)))
    y = frozen_bn(sd, pre + "bn3.", F.conv2d(y, sd[pre + "conv3.weight"]))
    return F.relu(y + idt)


RESNET50_BLOCKS = (3, 4, 6, 3)


def resnet_backbone(sd, cfg: OracleConfig, frames: Tensor, pre="backbone.0.body.") -> List[Tensor]:
    """frames [T,3,H,W] -> layer1..layer4 maps (backbone.py:64-74: strides 4, 8, 16, 32)."""
    x = F.relu(frozen_bn(sd, pre + "bn1.", F.conv2d(frames, sd[pre + "conv1.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    outs = []
    for li, blocks in enumerate(RESNET50_BLOCKS):
        for b in range(blocks):
            x = resnet_bottleneck(sd, f"{pre}layer{li + 1}.{b}.", x, 2 if (b == 0 and li > 0) else 1)
        outs.append(x)
    return outs


# --------------------------------------------------------------------------------------
# Video-Swin backbone (models/video_swin_transformer.py)
# --------------------------------------------------------------------------------------
def rel_pos_index_3d(wd: int, wh: int, ww: int) -> Tensor:
    """video_swin_transformer.py:114-128."""
    coords = torch.stack(torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing="ij"))
    cf = coords.flatten(1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += wd - 1
    rel[:, :, 1] += wh - 1
    rel[:, :, 2] += ww - 1
    rel[:, :, 0] *= (2 * wh - 1) * (2 * ww - 1)
    rel[:, :, 1] *= (2 * ww - 1)
    return rel.sum(-1)


def get_window_size_3d(x_size, window_size, shift_size):
    """video_swin_transformer.py:71-84."""
    use_w = list(window_size)
    use_s = list(shift_size)
    for i in range(3):
        if x_size[i] <= window_size[i]:
            use_w[i] = x_size[i]
            use_s[i] = 0
    return tuple(use_w), tuple(use_s)


def window_partition_3d(x, ws):
    B, D, H, W, C = x.shape
    x = x.view(B, D // ws[0], ws[0], H // ws[1], ws[1], W // ws[2], ws[2], C)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().view(-1, ws[0] * ws[1] * ws[2], C)


def window_reverse_3d(windows, ws, B, D, H, W):
    x = windows.view(B, D // ws[0], H // ws[1], W // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous().view(B, D, H, W, -1)


def compute_mask_3d(Dp, Hp, Wp, ws, ss):
    """video_swin_transformer.py:316-329."""
    img = torch.zeros(1, Dp, Hp, Wp, 1)
    cnt = 0
    for d in (slice(-ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
        for h in (slice(-ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
            for w in (slice(-ws[2]), slice(-ws[2], -ss[2]), slice(-ss[2], None)):
                img[:, d, h, w, :] = cnt
                cnt += 1
    mw = window_partition_3d(img, ws).squeeze(-1)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(am != 0, torch.tensor(-100.0), torch.tensor(0.0))


def window_attention_3d(sd, pre, xw, nh, full_ws, mask):
    """WindowAttention3D.forward video_swin_transformer.py:138-169."""
    B_, N, C = xw.shape
    d = C // nh
    qkv = linear(xw, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(B_, N, 3, nh, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (float(d) ** -0.5), qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    idx = rel_pos_index_3d(*full_ws)[:N, :N].reshape(-1)
    bias = sd[pre + "relative_position_bias_table"][idx].reshape(N, N, nh).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, nh, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, nh, N, N)
    attn = torch.softmax(attn, -1)
    x = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return linear(x, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def video_swin_block(sd, pre, x, nh, window_size, shift_size, mask_matrix):
    """SwinTransformerBlock3D.forward video_swin_transformer.py:215-274.  x [B,D,H,W,C]."""
    B, D, H, W, C = x.shape
    ws, ss = get_window_size_3d((D, H, W), window_size, shift_size)
    shortcut = x
    x = layer_norm(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    pad_d1 = (ws[0] - D % ws[0]) % ws[0]
    pad_b = (ws[1] - H % ws[1]) % ws[1]
    pad_r = (ws[2] - W % ws[2]) % ws[2]
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b, 0, pad_d1))
    _, Dp, Hp, Wp, _ = x.shape
    if any(i > 0 for i in ss):
        x = torch.roll(x, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
        am = mask_matrix
    else:
        am = None
    xw = window_partition_3d(x, ws)
    aw = window_attention_3d(sd, pre + "attn.", xw, nh, window_size, am)
    x = window_reverse_3d(aw, ws, B, Dp, Hp, Wp)
    if any(i > 0 for i in ss):
        x = torch.roll(x, shifts=(ss[0], ss[1], ss[2]), dims=(1, 2, 3))
    x = x[:, :D, :H, :W, :]
    x = shortcut + x
    y = layer_norm(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    y = linear(y, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])
    y = F.gelu(y)
    y = linear(y, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    return x + y


def video_patch_merging(sd, pre, x):
    """PatchMerging.forward video_swin_transformer.py:290-312.  x [B,D,H,W,C]."""
    B, D, H, W, C = x.shape
    if H % 2 == 1 or W % 2 == 1:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
    x = layer_norm(x, sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    return linear(x, sd[pre + "reduction.weight"])


def video_swin_backbone(sd, cfg: OracleConfig, clip: Tensor, pre="backbone.0.body.") -> List[Tensor]:
    """VideoSwinTransformerBackbone.forward video_swin_transformer.py:678-697.
    clip [B,T,3,H,W] -> 4 maps [(B T),Ci,hi,wi]; stage outputs taken BEFORE the merge, no output norm."""
    B, T = clip.shape[:2]
    x = clip.permute(0, 2, 1, 3, 4)  # b c t h w
    _, _, D, H0, W0 = x.shape
    if W0 % 4:
        x = F.pad(x, (0, 4 - W0 % 4))
    if H0 % 4:
        x = F.pad(x, (0, 0, 0, 4 - H0 % 4))
    x = F.conv3d(x, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"], stride=(1, 4, 4))
    _, C, D, Wh, Ww = x.shape
    x = x.flatten(2).transpose(1, 2)
    x = layer_norm(x, sd[pre + "patch_embed.norm.weight"], sd[pre + "patch_embed.norm.bias"])
    x = x.transpose(1, 2).view(B, C, D, Wh, Ww)
    window_size = tuple(cfg.video_window)
    shift_size = tuple(i // 2 for i in window_size)
    outs = []
    for i, depth in enumerate(cfg.depths):
        x = x.permute(0, 2, 3, 4, 1)  # b d h w c
        _, D, H, W, C = x.shape
        ws, ss = get_window_size_3d((D, H, W), window_size, shift_size)
        Dp = int(math.ceil(D / ws[0])) * ws[0]
        Hp = int(math.ceil(H / ws[1])) * ws[1]
        Wp = int(math.ceil(W / ws[2])) * ws[2]
        am = compute_mask_3d(Dp, Hp, Wp, ws, ss)
        for j in range(depth):
            x = video_swin_block(sd, f"{pre}layers.{i}.blocks.{j}.", x, cfg.num_heads[i], window_size,
                                 (0, 0, 0) if j % 2 == 0 else shift_size, am)
        x_out = x.permute(0, 4, 1, 2, 3)  # b c d h w
        outs.append(x_out.permute(0, 2, 1, 3, 4).reshape(B * D, C, H, W).contiguous())
        if i < len(cfg.depths) - 1:
            x = video_patch_merging(sd, f"{pre}downsamples.{i}.", x)
        x = x.permute(0, 4, 1, 2, 3)
    return outs


# --------------------------------------------------------------------------------------
# transformer (models/tce_deformable_transformer.py)
# --------------------------------------------------------------------------------------
def _mha_sd(sd, pre, q, k, v, nheads, kpm=None):
    return mha(q, k, v, sd[pre + "in_proj_weight"], sd[pre + "in_proj_bias"],
               sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"], nheads, kpm)


def _ffn(sd, pre, x, l1="linear1", l2="linear2"):
    y = linear(x, sd[pre + l1 + ".weight"], sd[pre + l1 + ".bias"])
    y = F.relu(y)
    return linear(y, sd[pre + l2 + ".weight"], sd[pre + l2 + ".bias"])


def _ln(sd, pre, x, eps=1e-5):
    return layer_norm(x, sd[pre + ".weight"], sd[pre + ".bias"], eps)


def frame_token_layer(sd, pre, cfg, src, pos, token, token_pos, shapes, padding_mask, valid_ratios):
    """FrameTokenLayer.forward tce_deformable_transformer.py:443-493."""
    B = token.shape[0]
    M, L, P = cfg.nheads, cfg.num_feature_levels, cfg.enc_n_points
    ref = torch.sigmoid(linear(token, sd[pre + "reference_points.weight"], sd[pre + "reference_points.bias"]))
    ref = ref[:, :, None] * valid_ratios[:, None]
    t2, _, _ = msda_module(sd, pre + "token_frame_atten.", token + token_pos, ref, src, shapes, padding_mask, M, L, P)
    token = _ln(sd, pre + "norm1", token + t2)
    # all T*F tokens of the clip as ONE sequence (batch axis = 1)
    tk = token.reshape(-1, token.shape[-1]).unsqueeze(1)
    tp = token_pos.reshape(-1, token.shape[-1]).unsqueeze(1)
    t2 = _mha_sd(sd, pre + "token_self_atten.", tk + tp, tk + tp, tk, M)
    tk = _ln(sd, pre + "norm2", tk + t2)
    token = tk.squeeze(1).view(B, -1, token.shape[-1])
    # pixels <- tokens of their own frame: (L = S, N = B*T)
    q = (src + pos).transpose(0, 1)
    k = (token + token_pos).transpose(0, 1)
    s2 = _mha_sd(sd, pre + "frame_token_atten.", q, k, token.transpose(0, 1), M).transpose(0, 1)
    src = _ln(sd, pre + "norm3", src + s2)
    src = _ln(sd, pre + "norm4", src + _ffn(sd, pre, src))
    return src, token


def encoder_reference_points(shapes, valid_ratios):
    """DeformableTransformerEncoder.get_reference_points :572-589."""
    refs = []
    for lvl, (H, W) in enumerate(shapes):
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=torch.float32),
                                torch.linspace(0.5, W - 0.5, W, dtype=torch.float32), indexing="ij")
        ry = ry.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
        rx = rx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
        refs.append(torch.stack((rx, ry), -1))
    ref = torch.cat(refs, 1)
    return ref[:, :, None] * valid_ratios[:, None]


def encoder(sd, cfg, src, shapes, valid_ratios, pos, padding_mask, pre="transformer.encoder."):
    """DeformableTransformerEncoder.forward :611-627 + layer :535-553."""
    M, L, P = cfg.nheads, cfg.num_feature_levels, cfg.enc_n_points
    ref = encoder_reference_points(shapes, valid_ratios)
    out = src
    N = src.shape[0]
    if cfg.f_token > 0:
        bus = sd[pre + "memory_bus"][None].repeat(N, 1, 1)
        bus_pos = sd[pre + "memory_pos"][None].repeat(N, 1, 1)
    for i in range(cfg.enc_layers):
        lp = f"{pre}layers.{i}."
        out, bus = encoder_layer(sd, lp, cfg, out, pos, ref, shapes, valid_ratios, padding_mask,
                                 bus if cfg.f_token > 0 else None, bus_pos if cfg.f_token > 0 else None)
    return out


def _mlp(sd, pre, x, n):
    for i in range(n):
        x = linear(x, sd[f"{pre}layers.{i}.weight"], sd[f"{pre}layers.{i}.bias"])
        if i < n - 1:
            x = F.relu(x)
    return x


def decoder_layer(sd, lp, cfg, out, query_pos, ref_in, memory, shapes, padding_mask):
    """DeformableTransformerDecoderLayer.forward tce_deformable_transformer.py:675-699 (IQT branch :683).  out, query_pos
    [B*T, Q, C]; ref_in [B*T, Q, L, 2 or 4] (already scaled by the valid ratios).  Returns (out, sampling locations, weights)."""
    M, L, P = cfg.nheads, cfg.num_feature_levels, cfg.dec_n_points
    q = out + query_pos
    if cfg.qtrans:
        # [B*T, Q, C] fed seq-first untransposed: sequence axis = frames, batch axis = query slots
        t2 = _mha_sd(sd, lp + "self_attn.", q, q, out, M)
    else:
        t2 = _mha_sd(sd, lp + "self_attn.", q.transpose(0, 1), q.transpose(0, 1), out.transpose(0, 1), M).transpose(0, 1)
    out = _ln(sd, lp + "norm2", out + t2)
    t2, loc, aw = msda_module(sd, lp + "cross_attn.", out + query_pos, ref_in, memory, shapes, padding_mask, M, L, P)
    out = _ln(sd, lp + "norm1", out + t2)
    out = _ln(sd, lp + "norm3", out + _ffn(sd, lp, out))
    return out, loc, aw


def encoder_layer(sd, lp, cfg, out, pos, ref, shapes, valid_ratios, padding_mask, bus=None, bus_pos=None):
    """DeformableTransformerEncoderLayer.forward :535-553: frame-token layer (f_token > 0) -> MSDA self-attention -> FFN."""
    M, L, P = cfg.nheads, cfg.num_feature_levels, cfg.enc_n_points
    if cfg.f_token > 0:
        out, bus = frame_token_layer(sd, lp + "ftoken_layers.", cfg, out, pos, bus, bus_pos, shapes, padding_mask, valid_ratios)
    s2, _, _ = msda_module(sd, lp + "self_attn.", out + pos, ref, out, shapes, padding_mask, M, L, P)
    out = _ln(sd, lp + "norm1", out + s2)
    out = _ln(sd, lp + "norm2", out + _ffn(sd, lp, out))
    return out, bus


def decoder(sd, cfg, tgt, ref, memory, shapes, valid_ratios, query_pos, padding_mask, pre="transformer.decoder."):
    """DeformableTransformerDecoder.forward :721-790 + layer :675-699 (IQT branch :683)."""
    M, L, P = cfg.nheads, cfg.num_feature_levels, cfg.dec_n_points
    out = tgt
    inter, inter_ref, inter_samples = [], [], []
    for lid in range(cfg.dec_layers):
        lp = f"{pre}layers.{lid}."
        if ref.shape[-1] == 4:
            ref_in = ref[:, :, None] * torch.cat([valid_ratios, valid_ratios], -1)[:, None]
        else:
            ref_in = ref[:, :, None] * valid_ratios[:, None]
        out, loc, aw = decoder_layer(sd, lp, cfg, out, query_pos, ref_in, memory, shapes, padding_mask)
        # top-30 sample bookkeeping :752-758
        N, Lq = loc.shape[:2]
        loc_n = loc / valid_ratios[:, None, None, :, None, :]
        wf = aw.reshape(N, Lq, -1)
        sf = loc_n.reshape(N, Lq, -1, 2)
        k = min(30, wf.shape[2])
        _, top_idx = wf.topk(k, dim=2)
        samples_keep = torch.gather(sf, 2, top_idx.unsqueeze(-1).repeat(1, 1, 1, 2))
        if cfg.with_box_refine:
            tmp = _mlp(sd, f"{pre}bbox_embed.{lid}.", out, 3)
            if ref.shape[-1] == 4:
                new_ref = torch.sigmoid(tmp + inverse_sigmoid(ref))
            else:
                tmp = tmp.clone()
                tmp[..., :2] = tmp[..., :2] + inverse_sigmoid(ref)
                new_ref = torch.sigmoid(tmp)
            ref = new_ref
        inter.append(out)
        inter_ref.append(ref)
        inter_samples.append(samples_keep)
    return torch.stack(inter), torch.stack(inter_ref), torch.stack(inter_samples)


def valid_ratio(mask: Tensor) -> Tensor:
    """get_valid_ratio :125-132.  mask [N,H,W] -> [N,2] (w,h)."""
    _, H, W = mask.shape
    vh = torch.sum(~mask[:, :, 0], 1).float() / H
    vw = torch.sum(~mask[:, 0, :], 1).float() / W
    return torch.stack([vw, vh], -1)


def deformable_transformer(sd, cfg, srcs, tgt, masks, poses, query_embed, pre="transformer."):
    """DeformableTransformer.forward :134-242 (two_stage off)."""
    src_f, mask_f, pos_f, shapes = [], [], [], []
    for lvl, (s, m, p) in enumerate(zip(srcs, masks, poses)):
        n, c, h, w = s.shape
        shapes.append((h, w))
        src_f.append(s.flatten(2).transpose(1, 2))
        mask_f.append(m.flatten(1))
        pos_f.append(p.flatten(2).transpose(1, 2) + sd[pre + "level_embed"][lvl].view(1, 1, -1))
    src_f, mask_f, pos_f = torch.cat(src_f, 1), torch.cat(mask_f, 1), torch.cat(pos_f, 1)
    vr = torch.stack([valid_ratio(m) for m in masks], 1)
    memory = encoder(sd, cfg, src_f, shapes, vr, pos_f, mask_f)
    bt, q, c = tgt.shape
    qe = query_embed.unsqueeze(0).expand(bt, -1, -1)
    ref = torch.sigmoid(linear(qe, sd[pre + "reference_points.weight"], sd[pre + "reference_points.bias"]))
    init_ref = ref
    hs, inter_ref, inter_samples = decoder(sd, cfg, tgt, ref, memory, shapes, vr, qe, mask_f)
    mem_feats = []
    idx = 0
    for lvl in range(cfg.num_feature_levels - 1):
        h, w = shapes[lvl]
        mem_feats.append(memory[:, idx:idx + h * w].reshape(bt, h, w, c).permute(0, 3, 1, 2).contiguous())
        idx += h * w
    return hs, mem_feats, init_ref, inter_ref, memory, inter_samples


# --------------------------------------------------------------------------------------
# pixel decoder (models/segmentation.py)
# --------------------------------------------------------------------------------------
def conv_gn(sd, pre, x, groups=8, relu=False, stride=1, padding=0, norm_key="norm"):
    x = F.conv2d(x, sd[pre + "weight"], sd.get(pre + "bias"), stride=stride, padding=padding)
    x = F.group_norm(x, groups, sd[pre + norm_key + ".weight"], sd[pre + norm_key + ".bias"], 1e-5)
    return F.relu(x) if relu else x


def vl_block(sd, pre, cfg, tgt, text, t, h, w, tgt_kpm, text_kpm, text_pos, query_pos, sr):
    """VisionLanguageBlock.forward_post segmentation.py:326-377.  tgt [(t h w), b, c]."""
    b = tgt.shape[1]
    nh = 8
    q = k = tgt + query_pos

    def to_map(z):
        return z.view(t, h, w, b, -1).permute(3, 0, 4, 1, 2).reshape(b * t, -1, h, w)

    if sr > 1:
        nh_, nw_ = int(h * 1.0 / sr), int(w * 1.0 / sr)
        qm = interp_nearest(to_map(q), (nh_, nw_))
        vm = interp_nearest(to_map(tgt), (nh_, nw_))

        def to_seq(z, hh, ww):
            return z.view(b, t, -1, hh, ww).permute(1, 3, 4, 0, 2).reshape(t * hh * ww, b, -1)

        q = k = to_seq(qm, nh_, nw_)
        v = to_seq(vm, nh_, nw_)
        kpm = interp_nearest(tgt_kpm.reshape(b * t, h, w).float(), (nh_, nw_)).bool().reshape(b, t, nh_, nw_).flatten(1)
    else:
        v, kpm = tgt, tgt_kpm
    tgt2 = _mha_sd(sd, pre + "self_attn.", q, k, v, nh, kpm)
    if sr > 1:
        m2 = tgt2.view(t, nh_, nw_, b, -1).permute(3, 0, 4, 1, 2).reshape(b * t, -1, nh_, nw_)
        m2 = interp_bilinear(m2, (h, w))
        tgt2 = m2.view(b, t, -1, h, w).permute(1, 3, 4, 0, 2).reshape(t * h * w, b, -1)
    tgt = _ln(sd, pre + "norm1", tgt + tgt2)
    tgt2 = _mha_sd(sd, pre + "multihead_attn.", tgt + query_pos, text + text_pos, text, nh, text_kpm)
    tgt = _ln(sd, pre + "norm2", tgt + tgt2)
    tgt = _ln(sd, pre + "norm3", tgt + _ffn(sd, pre, tgt))
    return tgt


def pixel_decoder(sd, cfg, feats, feat_masks, text_feat, text_mask, poses, memory, nf, pre="pixel_decoder."):
    """CrossModalFPNDecoder.forward segmentation.py:175-243,275-296.
    feats: 4 backbone maps (res2..res5); memory: 3 encoder maps (8x..32x); poses: 4 backbone pos maps."""
    text_pos = pos_sine_1d(text_mask, cfg.hidden_dim).permute(2, 0, 1)
    text = text_feat.permute(1, 0, 2)
    levels = [(memory[2], feat_masks[3], poses[3], 4), (memory[1], feat_masks[2], poses[2], 3),
              (memory[0], feat_masks[1], poses[1], 2), (feats[0], feat_masks[0], poses[0], 1)]
    sr_by_stage = {1: 8, 2: 4, 3: 2, 4: 1}
    y = None
    for x, xmask, pos, stage in levels:
        n, c, h, w = pos.shape
        b, t = n // nf, nf
        vis = conv_gn(sd, f"{pre}adapter_{stage}.", x, 8)
        if cfg.vlblock:
            seq = vis.view(b, t, -1, h, w).permute(1, 3, 4, 0, 2).reshape(t * h * w, b, -1)
            vpos = pos.view(b, t, -1, h, w).permute(1, 3, 4, 0, 2).reshape(t * h * w, b, -1)
            vmask = xmask.view(b, t * h * w)
            cur = vl_block(sd, f"{pre}cross_attn_{stage}.", cfg, seq, text, t, h, w, vmask, text_mask, text_pos, vpos,
                           sr_by_stage[stage])
            cur = cur.view(t, h, w, b, -1).permute(3, 0, 4, 1, 2).reshape(b * t, -1, h, w)
        else:
            cur = vis
        if y is None:
            y = conv_gn(sd, f"{pre}layer_{stage}.", cur, 8, relu=True, padding=1)
        else:
            y = cur + interp_nearest(y, (h, w))
            y = conv_gn(sd, f"{pre}layer_{stage}.", y, 8, relu=True, padding=1)
    return F.conv2d(y, sd[pre + "mask_features.weight"], sd[pre + "mask_features.bias"], padding=1)


# --------------------------------------------------------------------------------------
# mask head (models/tce_rvos.py:426-599)
# --------------------------------------------------------------------------------------
def dynamic_mask_head(cfg, mask_features, params, ref_xy, img_hw, stride=4):
    """mask_features [t,c,h,w]; params [t*q, n]; ref_xy [t*q, 2] (cx,cy normalised); -> [t*q, h, w]."""
    t, c, h, w = mask_features.shape
    nq = params.shape[0]
    q = nq // t
    ch = cfg.dynamic_mask_channels
    img_h, img_w = img_hw
    ref = ref_xy * torch.tensor([float(img_w), float(img_h)])
    sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32) + stride // 2
    sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32) + stride // 2
    feats = mask_features.unsqueeze(1).expand(t, q, c, h, w)
    if cfg.rel_coord:
        rel_x = ref[:, 0].view(t, q, 1, 1) - sx.view(1, 1, 1, w).expand(t, q, h, w)
        rel_y = ref[:, 1].view(t, q, 1, 1) - sy.view(1, 1, h, 1).expand(t, q, h, w)
        feats = torch.cat([feats, rel_x.unsqueeze(2), rel_y.unsqueeze(2)], dim=2)
    cin = feats.shape[2]
    x = feats.reshape(t * q, cin, h * w)
    # parameter layout [w0 | w1 | w2 | b0 | b1 | b2]  (parse_dynamic_params :536-559)
    n_layers = cfg.controller_layers
    wn, bn = [], []
    for l in range(n_layers):
        if l == 0:
            wn.append(cin * ch)
            bn.append(ch)
        elif l == n_layers - 1:
            wn.append(ch)
            bn.append(1)
        else:
            wn.append(ch * ch)
            bn.append(ch)
    splits = list(torch.split_with_sizes(params, wn + bn, dim=1))
    for l in range(n_layers):
        out_c = ch if l < n_layers - 1 else 1
        wl = splits[l].reshape(nq, out_c, -1)
        bl = splits[n_layers + l].reshape(nq, out_c, 1)
        x = torch.bmm(wl, x) + bl
        if l < n_layers - 1:
            x = F.relu(x)
    return x.reshape(nq, h, w)


# --------------------------------------------------------------------------------------
# full forward (models/tce_rvos.py:194-393)
# --------------------------------------------------------------------------------------
def text_resize(sd, x):
    """FeatureResizer tce_rvos.py:616-635 (LayerNorm eps 1e-12)."""
    return layer_norm(linear(x, sd["resizer.fc.weight"], sd["resizer.fc.bias"]),
                      sd["resizer.layer_norm.weight"], sd["resizer.layer_norm.bias"], 1e-12)


def fusion(sd, tgt, text, text_mask, text_pos, pre="fusion_module.multihead_attn."):
    """VisionLanguageFusionModule.forward segmentation.py:455-464: tgt * MHA(tgt, text+pos, text).
    nhead is the literal 8 of tce_rvos.py:153, independent of --nheads."""
    return tgt * _mha_sd(sd, pre, tgt, text + text_pos, text, 8, text_mask)


def forward(sd: Dict[str, Tensor], cfg: OracleConfig, frames: Tensor, text_hidden: Tensor, text_pooled: Tensor,
            text_attn_mask: Optional[Tensor] = None, img_size: Optional[Tuple[int, int]] = None,
            pad_mask: Optional[Tensor] = None, return_stages: bool = False, valid_index: Optional[int] = None):
    """One clip.  frames [T,3,H,W]; text_hidden [1,L,768]; text_pooled [1,768]; text_attn_mask [1,L] (1 = token).
    valid_index: targets[0]['valid_indices'] of the A2D / JHMDB single-frame path (tce_rvos.py:233-243): the backbone sees
    all T frames, everything after it only frame `valid_index` (t -> 1).
    Returns the reference's output dict (B = 1)."""
    sd = {k: v.detach().to(torch.float32).cpu() for k, v in sd.items() if not k.startswith("text_encoder.")}
    T, _, H, W = frames.shape
    b, t = 1, T
    if pad_mask is None:
        pad_mask = torch.zeros(T, H, W, dtype=torch.bool)
    if img_size is None:
        img_size = (H, W)
    Ltxt = text_hidden.shape[1]
    if text_attn_mask is None:
        text_attn_mask = torch.ones(1, Ltxt, dtype=torch.long)
    stages = {}

    # backbone + per-level masks + sine pos  (Joiner, swin_transformer.py:665-677,632-640)
    if cfg.is_resnet:
        feats = resnet_backbone(sd, cfg, frames)
    elif cfg.is_video_swin:
        feats = video_swin_backbone(sd, cfg, frames[None])
    else:
        feats = swin_backbone(sd, cfg, frames)
    feat_masks = [interp_nearest(pad_mask[None].float(), f.shape[-2:]).bool()[0] for f in feats]
    poses = [pos_sine_2d(m, cfg.hidden_dim // 2) for m in feat_masks]
    stages["backbone"] = feats
    if valid_index is not None:  # tce_rvos.py:233-243: index_select of features, masks, position maps and samples.mask; t -> 1
        vi = int(valid_index)
        feats = [f[vi:vi + 1] for f in feats]
        feat_masks = [m[vi:vi + 1] for m in feat_masks]
        poses = [p[vi:vi + 1] for p in poses]
        pad_mask = pad_mask[vi:vi + 1]
        t = 1

    # text  (forward_text :406-424)
    text_mask = text_attn_mask.ne(1).bool()
    text_feat = text_resize(sd, text_hidden)          # [1, L, C]
    text_sent = text_resize(sd, text_pooled)          # [1, C]
    text_pos = pos_sine_1d(text_mask, cfg.hidden_dim).permute(2, 0, 1)  # [L,1,C]
    text_word = text_feat.permute(1, 0, 2)            # [L,1,C]

    # input_proj + early fusion (:258-307)
    srcs, masks, pposes = [], [], []
    for l in range(3):
        f = feats[1 + l]
        s = F.conv2d(f, sd[f"input_proj.{l}.0.weight"], sd[f"input_proj.{l}.0.bias"])
        s = F.group_norm(s, 32, sd[f"input_proj.{l}.1.weight"], sd[f"input_proj.{l}.1.bias"], 1e-5)
        n, c, h, w = s.shape
        seq = s.view(b, t, c, h, w).permute(1, 3, 4, 0, 2).reshape(t * h * w, b, c)
        seq = fusion(sd, seq, text_word, text_mask, text_pos)
        s = seq.view(t, h, w, b, c).permute(3, 0, 4, 1, 2).reshape(b * t, c, h, w)
        srcs.append(s)
        masks.append(feat_masks[1 + l])
        pposes.append(poses[1 + l])
    for l in range(3, cfg.num_feature_levels):
        inp = feats[-1] if l == 3 else srcs[-1]
        s = F.conv2d(inp, sd[f"input_proj.{l}.0.weight"], sd[f"input_proj.{l}.0.bias"], stride=2, padding=1)
        s = F.group_norm(s, 32, sd[f"input_proj.{l}.1.weight"], sd[f"input_proj.{l}.1.bias"], 1e-5)
        n, c, h, w = s.shape
        m = interp_nearest(pad_mask[None].float(), (h, w)).bool()[0]
        p = pos_sine_2d(m, cfg.hidden_dim // 2)
        seq = s.view(b, t, c, h, w).permute(1, 3, 4, 0, 2).reshape(t * h * w, b, c)
        seq = fusion(sd, seq, text_word, text_mask, text_pos)
        s = seq.view(t, h, w, b, c).permute(3, 0, 4, 1, 2).reshape(b * t, c, h, w)
        srcs.append(s)
        masks.append(m)
        pposes.append(p)
    stages["srcs"] = srcs

    # transformer
    tgt = text_sent[:, None, None, :].expand(b, t, cfg.num_queries, -1).reshape(b * t, cfg.num_queries, -1)
    hs, mem_feats, init_ref, inter_ref, memory, inter_samples = deformable_transformer(
        sd, cfg, srcs, tgt, masks, pposes, sd["query_embed.weight"])
    stages["memory"] = memory
    stages["hs"] = hs

    # heads (:330-365)
    classes, coords, visibles = [], [], []
    for lvl in range(hs.shape[0]):
        reference = inverse_sigmoid(init_ref if lvl == 0 else inter_ref[lvl - 1])
        ci = lvl if cfg.with_box_refine else 0
        oc = linear(hs[lvl], sd[f"class_embed.{ci}.weight"], sd[f"class_embed.{ci}.bias"])
        if cfg.vis_loss:
            visibles.append(linear(hs[lvl], sd[f"visible_embed.{ci}.weight"], sd[f"visible_embed.{ci}.bias"]))
        tmp = _mlp(sd, f"bbox_embed.{ci}.", hs[lvl], 3)
        if reference.shape[-1] == 4:
            tmp = tmp + reference
        else:
            tmp = tmp.clone()
            tmp[..., :2] = tmp[..., :2] + reference
        classes.append(oc)
        coords.append(torch.sigmoid(tmp))
    out = {"pred_logits": classes[-1].view(b, t, cfg.num_queries, -1),
           "pred_boxes": coords[-1].view(b, t, cfg.num_queries, 4)}
    if cfg.vis_loss:
        out["pred_visible"] = visibles[-1].view(b, t, cfg.num_queries, 1)
    if cfg.contrastive:  # contrastive_cal :512-521 ("enc_outputs_class" of the non-two-stage transformer IS the encoder memory)
        vis_mem = memory.view(b, t, memory.shape[1], -1).mean(2)
        out["contrastive"] = F.cosine_similarity(vis_mem, text_sent.view(b, 1, -1).repeat(1, t, 1), dim=2, eps=1e-6)

    # pixel decoder + dynamic conv (:367-380)
    mask_features = pixel_decoder(sd, cfg, feats, feat_masks, text_feat, text_mask, poses, mem_feats, t)
    stages["mask_features"] = mask_features
    seg = []
    for lvl in range(hs.shape[0]):
        params = _mlp(sd, "controller.", hs[lvl], 3).reshape(b * t * cfg.num_queries, -1)
        refs = inter_ref[lvl, ..., :2].reshape(b * t * cfg.num_queries, 2)
        m = dynamic_mask_head(cfg, mask_features, params, refs, img_size)
        seg.append(m.view(b, t, cfg.num_queries, m.shape[-2], m.shape[-1]))
    out["pred_masks"] = seg[-1]
    if cfg.aux_loss:
        out["aux_outputs"] = [{"pred_logits": classes[i].view(b, t, cfg.num_queries, -1),
                               "pred_boxes": coords[i].view(b, t, cfg.num_queries, 4),
                               "pred_masks": seg[i]} for i in range(len(seg) - 1)]
        if cfg.vis_loss:
            for i in range(len(seg) - 1):
                out["aux_outputs"][i]["pred_visible"] = visibles[i].view(b, t, cfg.num_queries, 1)
    out["reference_points"] = inter_ref[-2, :, :, :2].view(b, t, cfg.num_queries, 2)
    out["memory"] = memory
    if return_stages:
        out["_stages"] = stages
        out["_inter_samples"] = inter_samples
    return out


# --------------------------------------------------------------------------------------
# caller harness H (inference_ytvos.py:238-250) and the IoU metric (davis2017/metrics.py:6-37)
# --------------------------------------------------------------------------------------
def select_masks(pred_logits: Tensor, pred_masks: Tensor, out_hw: Tuple[int, int], threshold: float = 0.5):
    """pred_logits [t,q,k], pred_masks [t,q,h,w] -> (bool [t,H0,W0], best query index)."""
    scores = pred_logits.sigmoid().mean(0)      # [q,k]
    max_scores, _ = scores.max(-1)
    best = int(torch.argmax(max_scores))
    m = pred_masks[:, best]                     # [t,h,w]
    m = interp_bilinear(m[:, None], out_hw)[:, 0]
    return m.sigmoid() > threshold, best


def mask_iou(a: Tensor, b: Tensor) -> float:
    inter = (a & b).sum().item()
    union = (a | b).sum().item()
    return 1.0 if union == 0 else inter / union
