"""Model configuration derived from the reference's argparse namespace (opts.py:3-156) and the enumeration of
every parameter the reference model owns (names + shapes = the checkpoint contract)."""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

# swin_transformer.py:687-745, video_swin_transformer.py:733-779
BACKBONES = {
    "swin_t_p4w7": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), video=False),
    "swin_s_p4w7": dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24), video=False),
    "swin_b_p4w7": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), video=False),
    "swin_l_p4w7": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), video=False),
    "video_swin_t_p4w7": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), video=True),
    "video_swin_s_p4w7": dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24), video=True),
    "video_swin_b_p4w7": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), video=True),
    # models/backbone.py:59-101 (torchvision bottleneck ResNets; BASELINE config 1 = resnet50)
    "resnet50": dict(resnet_blocks=(3, 4, 6, 3)),
    "resnet101": dict(resnet_blocks=(3, 4, 23, 3)),
}


@dataclass
class ModelConfig:
    backbone: str = "swin_t_p4w7"
    embed_dim: int = 96
    depths: Tuple[int, ...] = (2, 2, 6, 2)
    num_heads: Tuple[int, ...] = (3, 6, 12, 24)
    video: bool = False
    resnet_blocks: Tuple[int, ...] = ()  # non-empty: bottleneck ResNet backbone instead of Swin
    window_size: int = 7
    video_window: Tuple[int, int, int] = (8, 7, 7)
    mlp_ratio: float = 4.0
    hidden_dim: int = 256
    nheads: int = 8
    num_feature_levels: int = 4
    enc_layers: int = 4
    dec_layers: int = 4
    dim_feedforward: int = 2048
    enc_n_points: int = 4
    dec_n_points: int = 4
    num_queries: int = 5
    num_frames: int = 5
    f_token: int = 8
    qtrans: bool = True
    with_box_refine: bool = True
    mask_dim: int = 256
    controller_layers: int = 3
    dynamic_mask_channels: int = 8
    rel_coord: bool = True
    vlblock: bool = True
    aux_loss: bool = True
    num_classes: int = 1
    text_hidden: int = 768
    vis_loss: bool = False      # --vis_loss: visible_embed heads -> out['pred_visible'] (tce_rvos.py:62-63,328-363)
    contrastive: bool = False   # --contrastive: out['contrastive'] = cos(mean_S memory, sentence feature) (:318-319,512-521)

    @property
    def is_resnet(self):
        return len(self.resnet_blocks) > 0

    @property
    def num_channels(self):
        if self.is_resnet:  # backbone.py:68
            return [256, 512, 1024, 2048]
        return [self.embed_dim * 2 ** i for i in range(len(self.depths))]

    @property
    def num_gen_params(self):
        c = self.dynamic_mask_channels
        cin = self.mask_dim + (2 if self.rel_coord else 0)
        return cin * c + c * c + c + c + c + 1


def config_from_args(args) -> ModelConfig:
    """Reads the reference's flat argparse namespace unchanged (tolerates the attributes the reference
    itself forgets to define, e.g. f_extra)."""
    name = getattr(args, "backbone", "swin_t_p4w7")
    if name not in BACKBONES:
        raise ValueError(f"backbone '{name}' is outside the MI355X hot path (supported: {sorted(BACKBONES)}); "
                         f"X3D is not a SURVEY.md section 8 row")
    b = dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), video=False, resnet_blocks=())
    b.update(BACKBONES[name])
    if b["resnet_blocks"] and getattr(args, "dilation", False):
        raise NotImplementedError("--dilation (DC5) is not part of any BASELINE config")
    if getattr(args, "two_stage", False):
        raise AssertionError("args.two_stage must be false!")  # tce_rvos.py:102
    if getattr(args, "binary", False):
        num_classes = 1
    else:
        ds = getattr(args, "dataset_file", "ytvos")
        num_classes = {"ytvos": 65, "davis": 78, "a2d": 1, "jhmdb": 1}.get(ds, 91)  # tce_rvos.py:639-649
    if getattr(args, "f_token", 0) < 0:
        raise NotImplementedError("f_token < 0 (LastLayerAsToken) is not on the hot path")
    if getattr(args, "controller_layers", 3) != 3 or getattr(args, "dynamic_mask_channels", 8) != 8:
        raise NotImplementedError("the mask-head kernels are built for controller_layers=3, dynamic_mask_channels=8")
    if not getattr(args, "rel_coord", True):
        raise NotImplementedError("--no_rel_coord is not supported by the mask-head kernel")
    return ModelConfig(
        backbone=name, embed_dim=b["embed_dim"], depths=tuple(b["depths"]), num_heads=tuple(b["num_heads"]),
        video=b["video"], resnet_blocks=tuple(b["resnet_blocks"]), hidden_dim=getattr(args, "hidden_dim", 256), nheads=getattr(args, "nheads", 8),
        num_feature_levels=getattr(args, "num_feature_levels", 4), enc_layers=getattr(args, "enc_layers", 4),
        dec_layers=getattr(args, "dec_layers", 4), dim_feedforward=getattr(args, "dim_feedforward", 2048),
        enc_n_points=getattr(args, "enc_n_points", 4), dec_n_points=getattr(args, "dec_n_points", 4),
        num_queries=getattr(args, "num_queries", 5), num_frames=getattr(args, "num_frames", 5),
        f_token=getattr(args, "f_token", 0), qtrans=bool(getattr(args, "qtrans", False)),
        with_box_refine=bool(getattr(args, "with_box_refine", False)), mask_dim=getattr(args, "mask_dim", 256),
        rel_coord=True, vlblock=bool(getattr(args, "vlblock", True)), aux_loss=bool(getattr(args, "aux_loss", True)),
        num_classes=num_classes, vis_loss=bool(getattr(args, "vis_loss", False)), contrastive=bool(getattr(args, "contrastive", False)))


def param_shapes(cfg: ModelConfig) -> "OrderedDict[str, tuple]":
    """{state-dict key: shape} of every float parameter (text_encoder.* excluded; it is the HF module's)."""
    S = OrderedDict()
    d, ff = cfg.hidden_dim, cfg.dim_feedforward
    M = cfg.nheads

    def lin(pre, o, i, bias=True):
        S[pre + ".weight"] = (o, i)
        if bias:
            S[pre + ".bias"] = (o,)

    def ln(pre, c):
        S[pre + ".weight"] = (c,)
        S[pre + ".bias"] = (c,)

    def mha(pre):
        S[pre + ".in_proj_weight"] = (3 * d, d)
        S[pre + ".in_proj_bias"] = (3 * d,)
        lin(pre + ".out_proj", d, d)

    def msda(pre, L, P):
        lin(pre + ".sampling_offsets", M * L * P * 2, d)
        lin(pre + ".attention_weights", M * L * P, d)
        lin(pre + ".value_proj", d, d)
        lin(pre + ".output_proj", d, d)

    L = cfg.num_feature_levels
    # transformer (registration order of the reference: transformer first)
    for i in range(cfg.enc_layers):
        p = f"transformer.encoder.layers.{i}"
        if cfg.f_token > 0:
            f = p + ".ftoken_layers"
            lin(f + ".reference_points", 2, d)
            msda(f + ".token_frame_atten", L, cfg.enc_n_points)
            ln(f + ".norm1", d)
            mha(f + ".token_self_atten")
            ln(f + ".norm2", d)
            mha(f + ".frame_token_atten")
            ln(f + ".norm3", d)
            lin(f + ".linear1", ff, d)
            lin(f + ".linear2", d, ff)
            ln(f + ".norm4", d)
        msda(p + ".self_attn", L, cfg.enc_n_points)
        ln(p + ".norm1", d)
        lin(p + ".linear1", ff, d)
        lin(p + ".linear2", d, ff)
        ln(p + ".norm2", d)
    if cfg.f_token > 0:
        S["transformer.encoder.memory_bus"] = (cfg.f_token, 256)  # d_model literal of the reference (:557,562)
        S["transformer.encoder.memory_pos"] = (cfg.f_token, 256)
    for i in range(cfg.dec_layers):
        p = f"transformer.decoder.layers.{i}"
        msda(p + ".cross_attn", L, cfg.dec_n_points)
        ln(p + ".norm1", d)
        mha(p + ".self_attn")
        ln(p + ".norm2", d)
        lin(p + ".linear1", ff, d)
        lin(p + ".linear2", d, ff)
        ln(p + ".norm3", d)
    S["transformer.level_embed"] = (L, d)
    lin("transformer.reference_points", 2, d)
    n_pred = cfg.dec_layers if cfg.with_box_refine else 1
    for i in range(n_pred):
        lin(f"class_embed.{i}", cfg.num_classes, d)
    if cfg.vis_loss:  # tce_rvos.py:62-63,119-120
        for i in range(n_pred):
            lin(f"visible_embed.{i}", 1, d)
    for i in range(n_pred):
        lin(f"bbox_embed.{i}.layers.0", d, d)
        lin(f"bbox_embed.{i}.layers.1", d, d)
        lin(f"bbox_embed.{i}.layers.2", 4, d)
    S["query_embed.weight"] = (cfg.num_queries, d)
    ch = cfg.num_channels
    for l in range(3):
        S[f"input_proj.{l}.0.weight"] = (d, ch[1 + l], 1, 1)
        S[f"input_proj.{l}.0.bias"] = (d,)
        ln(f"input_proj.{l}.1", d)
    for l in range(3, L):
        S[f"input_proj.{l}.0.weight"] = (d, ch[3] if l == 3 else d, 3, 3)
        S[f"input_proj.{l}.0.bias"] = (d,)
        ln(f"input_proj.{l}.1", d)
    # backbone
    b = "backbone.0.body"
    if cfg.is_resnet:
        _resnet_shapes(S, cfg, b)
    elif cfg.video:
        S[b + ".patch_embed.proj.weight"] = (cfg.embed_dim, 3, 1, 4, 4)
    else:
        S[b + ".patch_embed.proj.weight"] = (cfg.embed_dim, 3, 4, 4)
    if not cfg.is_resnet:
        S[b + ".patch_embed.proj.bias"] = (cfg.embed_dim,)
        ln(b + ".patch_embed.norm", cfg.embed_dim)
    ws = cfg.window_size
    for i, depth in enumerate(() if cfg.is_resnet else cfg.depths):
        c = ch[i]
        for j in range(depth):
            p = f"{b}.layers.{i}.blocks.{j}"
            ln(p + ".norm1", c)
            if cfg.video:
                wd = cfg.video_window
                S[p + ".attn.relative_position_bias_table"] = ((2 * wd[0] - 1) * (2 * wd[1] - 1) * (2 * wd[2] - 1),
                                                               cfg.num_heads[i])
            else:
                S[p + ".attn.relative_position_bias_table"] = ((2 * ws - 1) * (2 * ws - 1), cfg.num_heads[i])
            lin(p + ".attn.qkv", 3 * c, c)
            lin(p + ".attn.proj", c, c)
            ln(p + ".norm2", c)
            hid = int(c * cfg.mlp_ratio)
            lin(p + ".mlp.fc1", hid, c)
            lin(p + ".mlp.fc2", c, hid)
        if i < len(cfg.depths) - 1:
            p = f"{b}.downsamples.{i}" if cfg.video else f"{b}.layers.{i}.downsample"
            lin(p + ".reduction", 2 * c, 4 * c, bias=False)
            ln(p + ".norm", 4 * c)
    if not cfg.video and not cfg.is_resnet:
        for i in range(len(cfg.depths)):
            ln(f"{b}.norm{i}", ch[i])
    lin("resizer.fc", d, cfg.text_hidden)
    ln("resizer.layer_norm", d)
    mha("fusion_module.multihead_attn")
    fc = [ch[0], d, d, d]
    for k in range(1, 5):
        S[f"pixel_decoder.adapter_{k}.weight"] = (d, fc[k - 1], 1, 1)
        ln(f"pixel_decoder.adapter_{k}.norm", d)
        S[f"pixel_decoder.layer_{k}.weight"] = (d, d, 3, 3)
        ln(f"pixel_decoder.layer_{k}.norm", d)
    S["pixel_decoder.mask_features.weight"] = (cfg.mask_dim, d, 3, 3)
    S["pixel_decoder.mask_features.bias"] = (cfg.mask_dim,)
    if cfg.vlblock:
        for k in range(1, 5):
            p = f"pixel_decoder.cross_attn_{k}"
            mha(p + ".self_attn")
            mha(p + ".multihead_attn")
            lin(p + ".linear1", ff, d)
            lin(p + ".linear2", d, ff)
            ln(p + ".norm1", d)
            ln(p + ".norm2", d)
            ln(p + ".norm3", d)
    lin("controller.layers.0", d, d)
    lin("controller.layers.1", d, d)
    lin("controller.layers.2", cfg.num_gen_params, d)
    return S


FROZEN_BN_LEAVES = ("weight", "bias", "running_mean", "running_var")  # buffers, not parameters (backbone.py:31-34)


def _resnet_shapes(S, cfg, b):
    """torchvision bottleneck ResNet under IntermediateLayerGetter (backbone.py:64-74: conv1, bn1, layer1..4; fc and
    avgpool are dropped), every norm a FrozenBatchNorm2d."""
    def bn(pre, c):
        for leaf in FROZEN_BN_LEAVES:
            S[f"{pre}.{leaf}"] = (c,)

    S[b + ".conv1.weight"] = (64, 3, 7, 7)
    bn(b + ".bn1", 64)
    cin = 64
    for li, blocks in enumerate(cfg.resnet_blocks):
        width = 64 * 2 ** li
        for j in range(blocks):
            p = f"{b}.layer{li + 1}.{j}"
            S[p + ".conv1.weight"] = (width, cin, 1, 1)
            bn(p + ".bn1", width)
            S[p + ".conv2.weight"] = (width, width, 3, 3)
            bn(p + ".bn2", width)
            S[p + ".conv3.weight"] = (width * 4, width, 1, 1)
            bn(p + ".bn3", width * 4)
            if j == 0:
                S[p + ".downsample.0.weight"] = (width * 4, cin, 1, 1)
                bn(p + ".downsample.1", width * 4)
            cin = width * 4


def index_buffers(cfg: ModelConfig) -> "OrderedDict[str, tuple]":
    """Integer buffers the reference registers (relative_position_index) -- kept in the state dict for key
    compatibility; the kernels compute the index analytically."""
    B = OrderedDict()
    b = "backbone.0.body"
    for i, depth in enumerate(() if cfg.is_resnet else cfg.depths):
        for j in range(depth):
            if cfg.video:
                n = cfg.video_window[0] * cfg.video_window[1] * cfg.video_window[2]
            else:
                n = cfg.window_size ** 2
            B[f"{b}.layers.{i}.blocks.{j}.attn.relative_position_index"] = (n, n)
    return B
