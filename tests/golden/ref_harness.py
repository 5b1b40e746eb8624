"""Import harness for the upstream reference (CONTAINER ONLY -- never runs on the GPU box).

The reference tree at /root/reference is importable on CPU once four absent
packages are stood in for.  This module registers those
stand-ins, imports the reference's ``models`` package, and exposes helpers to
build a reference ``ReferFormer`` with a locally constructed (random-init)
RoBERTa and a fixed synthetic tokenisation.  It is used ONLY by
``make_golden.py`` (fixture generation).

Nothing here is shipped or imported by the product path, the tests, smoke() or
bench.py: the fixtures it produces are data (inputs + expected outputs).
"""
import os
import sys
import types
import math

REF_ROOT = os.environ.get("TCE_REFERENCE_ROOT", "/root/reference")


def _install_standins():
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    import transformers  # noqa: F401  (must be imported before the fake torchvision)

    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.__version__ = "0.15.0"
        tv.__path__ = []
        tv_models = types.ModuleType("torchvision.models")
        tv_models.__path__ = []
        tv_models_utils = types.ModuleType("torchvision.models._utils")

        class IntermediateLayerGetter(nn.ModuleDict):
            """torchvision is absent: the children of `model` up to the last requested one, run in order, the
            requested outputs collected under their new names (what backbone.py:74,77 relies on)."""

            def __init__(self, model, return_layers):
                todo = dict(return_layers)
                kept = {}
                for name, child in model.named_children():
                    kept[name] = child
                    todo.pop(name, None)
                    if not todo:
                        break
                super().__init__(kept)
                self.return_layers = dict(return_layers)

            def forward(self, x):
                out = {}
                for name, child in self.items():
                    x = child(x)
                    if name in self.return_layers:
                        out[self.return_layers[name]] = x
                return out

        tv_models_utils.IntermediateLayerGetter = IntermediateLayerGetter

        class _Bottleneck(nn.Module):
            # ResNet "v1.5" bottleneck as published (He et al. 2016; torchvision places the stride on the 3x3)
            def __init__(self, cin, width, stride, norm_layer, project):
                super().__init__()
                self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
                self.bn1 = norm_layer(width)
                self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
                self.bn2 = norm_layer(width)
                self.conv3 = nn.Conv2d(width, width * 4, 1, bias=False)
                self.bn3 = norm_layer(width * 4)
                self.relu = nn.ReLU(inplace=True)
                self.downsample = None
                if project:
                    self.downsample = nn.Sequential(nn.Conv2d(cin, width * 4, 1, stride=stride, bias=False),
                                                    norm_layer(width * 4))

            def forward(self, x):
                idt = x if self.downsample is None else self.downsample(x)
                y = self.relu(self.bn1(self.conv1(x)))
                y = self.relu(self.bn2(self.conv2(y)))
                y = self.bn3(self.conv3(y))
                return self.relu(y + idt)

        class _ResNet50(nn.Module):
            def __init__(self, norm_layer):
                super().__init__()
                self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
                self.bn1 = norm_layer(64)
                self.relu = nn.ReLU(inplace=True)
                self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
                cin = 64
                for li, (width, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
                    layer = []
                    for b in range(blocks):
                        layer.append(_Bottleneck(cin, width, stride if b == 0 else 1, norm_layer, b == 0))
                        cin = width * 4
                    setattr(self, f"layer{li + 1}", nn.Sequential(*layer))
                self.avgpool = nn.AdaptiveAvgPool2d(1)
                self.fc = nn.Linear(2048, 1000)

        def resnet50(replace_stride_with_dilation=None, pretrained=False, norm_layer=None, **kw):
            # `pretrained` would download weights (backbone.py:94-96): never honoured here, weights are synthetic
            assert not any(replace_stride_with_dilation or ()), "dilation is not part of any BASELINE config"
            return _ResNet50(norm_layer or nn.BatchNorm2d)

        tv_models.resnet50 = resnet50
        tv_ops = types.ModuleType("torchvision.ops")
        tv_ops.__path__ = []
        tv_ops_misc = types.ModuleType("torchvision.ops.misc")
        tv_ops_misc.interpolate = F.interpolate
        tv_ops_boxes = types.ModuleType("torchvision.ops.boxes")

        def box_area(boxes):
            return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])

        tv_ops_boxes.box_area = box_area
        tv_transforms = types.ModuleType("torchvision.transforms")
        tv_transforms.__path__ = []
        tv_tf = types.ModuleType("torchvision.transforms.functional")
        tv.models, tv.ops, tv.transforms = tv_models, tv_ops, tv_transforms
        tv_models._utils = tv_models_utils
        tv_ops.misc, tv_ops.boxes = tv_ops_misc, tv_ops_boxes
        tv_transforms.functional = tv_tf
        for name, mod in [("torchvision", tv), ("torchvision.models", tv_models),
                          ("torchvision.models._utils", tv_models_utils), ("torchvision.ops", tv_ops),
                          ("torchvision.ops.misc", tv_ops_misc), ("torchvision.ops.boxes", tv_ops_boxes),
                          ("torchvision.transforms", tv_transforms),
                          ("torchvision.transforms.functional", tv_tf)]:
            sys.modules[name] = mod

    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        timm.__path__ = []
        timm_models = types.ModuleType("timm.models")
        timm_models.__path__ = []
        timm_layers = types.ModuleType("timm.models.layers")

        class DropPath(nn.Module):  # identity in eval; the harness only runs eval
            def __init__(self, p=0.0):
                super().__init__()
                self.p = p

            def forward(self, x):
                assert not self.training or self.p == 0.0
                return x

        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

        def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
            return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)

        timm_layers.DropPath, timm_layers.to_2tuple, timm_layers.trunc_normal_ = DropPath, to_2tuple, trunc_normal_
        timm.models, timm_models.layers = timm_models, timm_layers
        sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.layers": timm_layers})

    if "fvcore" not in sys.modules:
        fv = types.ModuleType("fvcore")
        fv.__path__ = []
        fv_nn = types.ModuleType("fvcore.nn")
        fv_nn.__path__ = []
        fv_wi = types.ModuleType("fvcore.nn.weight_init")

        def c2_xavier_fill(module):
            nn.init.kaiming_uniform_(module.weight, a=1)
            if module.bias is not None:
                nn.init.constant_(module.bias, 0)

        fv_wi.c2_xavier_fill = c2_xavier_fill
        fv.nn, fv_nn.weight_init = fv_nn, fv_wi
        sys.modules.update({"fvcore": fv, "fvcore.nn": fv_nn, "fvcore.nn.weight_init": fv_wi})

    if "pycocotools" not in sys.modules:
        pc = types.ModuleType("pycocotools")
        pc.__path__ = []
        pcm = types.ModuleType("pycocotools.mask")
        pc.mask = pcm
        sys.modules.update({"pycocotools": pc, "pycocotools.mask": pcm})

    if "MultiScaleDeformableAttention_update" not in sys.modules:
        msda = types.ModuleType("MultiScaleDeformableAttention_update")

        def ms_deform_attn_forward(value, shapes, level_start, loc, weights, im2col_step, is_3d=False):
            # the reference's own pure-PyTorch core (ms_deform_attn_func.py:67-87)
            from models.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
            return ms_deform_attn_core_pytorch(value, shapes.tolist(), loc, weights)

        msda.ms_deform_attn_forward = ms_deform_attn_forward
        sys.modules["MultiScaleDeformableAttention_update"] = msda


_IMPORTED = False


def import_reference():
    """Returns the reference's ``models`` package (and friends) after installing stand-ins."""
    global _IMPORTED
    sys.dont_write_bytecode = True
    _install_standins()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import models.tce_rvos as ref_tce  # noqa
    _IMPORTED = True
    return ref_tce


class _FixedTokenizer:
    """Stands where RobertaTokenizerFast stands: returns fixed synthetic ids (BASELINE: 32-token text)."""

    def __init__(self, ids):
        self.ids = ids

    def batch_encode_plus(self, captions, padding="longest", return_tensors="pt"):
        import torch
        b = len(captions)
        ids = self.ids[:b] if self.ids.shape[0] >= b else self.ids.expand(b, -1)

        class _Enc(dict):
            def to(self, device):
                for k in list(self.keys()):
                    self[k] = self[k].to(device)
                return self

            def __getattr__(self, k):
                return self[k]

        return _Enc(input_ids=ids.clone(), attention_mask=torch.ones_like(ids))


def synthetic_token_ids(n_tokens=32, seed=1234, batch=1):
    import torch
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, 50265, (batch, n_tokens), generator=g)
    ids[:, 0] = 0
    ids[:, -1] = 2
    return ids


def reference_args(backbone="swin_t_p4w7", extra=()):
    sys.path.insert(0, REF_ROOT) if REF_ROOT not in sys.path else None
    import opts
    parser = opts.get_args_parser()
    argv = ["--with_box_refine", "--binary", "--freeze_text_encoder", "--f_token", "8", "--qtrans",
            "--backbone", backbone] + list(extra)
    args = parser.parse_args(argv)
    args.masks = True
    args.device = "cpu"
    return args


def build_reference_model(args, seed=0, roberta_layers=12, swin_cfg_override=None, token_ids=None):
    """Builds the reference ReferFormer on CPU with a random-init RoBERTa and a fixed tokenizer."""
    import torch
    import transformers
    ref_tce = import_reference()

    cfg = transformers.RobertaConfig(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1,
                                     pad_token_id=1, num_hidden_layers=roberta_layers)
    if token_ids is None:
        token_ids = synthetic_token_ids()

    class _Roberta:
        @staticmethod
        def from_pretrained(name):
            return transformers.RobertaModel(cfg)

    class _Tok:
        @staticmethod
        def from_pretrained(name):
            return _FixedTokenizer(token_ids)

    ref_tce.RobertaModel = _Roberta
    ref_tce.RobertaTokenizerFast = _Tok

    if swin_cfg_override is not None:
        import models.swin_transformer as ref_swin
        ref_swin.configs[args.backbone].update(swin_cfg_override)
        import models.video_swin_transformer as ref_vswin
        if args.backbone in ref_vswin.configs:
            ref_vswin.configs[args.backbone].update(swin_cfg_override)

    torch.manual_seed(seed)
    model, _crit, _post = ref_tce.build(args)
    model.eval()
    return model
