"""Deterministic synthetic weights, keyed by state-dict name.

There is no trained TCE-RVOS checkpoint (reference README: "Model Zoo: Coming Soon") and no network, so
every parity test, the golden fixtures and the benchmark run on synthetic weights.  A tensor's values
depend only on (its state-dict key, its shape, `salt`): the build container (where the reference runs and
fixtures are generated), the CPU oracle and the HIP path on the GPU box therefore all see the same numbers
without any weight file travelling.  Unlike the reference's initialisers (which zero the MSDA offset /
attention projections and the last bbox layer) every tensor here carries signal, so every path is exercised.

Pure torch-CPU, no HIP dependency.
"""
import math
import re
import zlib

import torch

_NORM_W = re.compile(r"(^|\.)(norm\d*|layer_norm|norm)\.weight$|input_proj\.\d+\.1\.weight$")
_NORM_B = re.compile(r"(^|\.)(norm\d*|layer_norm|norm)\.bias$|input_proj\.\d+\.1\.bias$")


_FBN = re.compile(r"^backbone\.0\.body\..*(bn\d|downsample\.1)\.(weight|bias|running_mean|running_var)$")


def _gen(key: str, salt: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (salt * 0x9E3779B1)) & 0x7FFFFFFF)


def synth_tensor(key: str, shape, salt: int = 0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    # the reference aliases transformer.decoder.bbox_embed to bbox_embed (tce_rvos.py:124): one tensor, two keys
    key = key.replace("transformer.decoder.bbox_embed.", "bbox_embed.")
    g = _gen(key, salt)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if _FBN.search(key):  # FrozenBatchNorm2d buffers of the ResNet backbone (backbone.py:20-56)
        if key.endswith("running_var"):
            return 0.6 + 0.8 * torch.rand(shape, generator=g, dtype=torch.float32)
        if key.endswith("running_mean"):
            return 0.1 * r
        return 1.0 + 0.1 * r if key.endswith("weight") else 0.05 * r
    if _NORM_W.search(key):
        return 1.0 + 0.1 * r
    if _NORM_B.search(key):
        return 0.05 * r
    if key.endswith("sampling_offsets.bias"):
        return 1.5 * r
    if key.endswith("relative_position_bias_table"):
        return 0.5 * r
    if key.endswith(("level_embed", "memory_bus", "memory_pos", "query_embed.weight")):
        return 0.5 * r
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return r / math.sqrt(fan_in)
    return 0.05 * r  # biases


def synth_state_dict(shapes: dict, salt: int = 0, skip_prefixes=("text_encoder.",)) -> dict:
    """shapes: {key: shape}.  Integer buffers (relative_position_index) are left to their owner."""
    out = {}
    for k, shp in shapes.items():
        if k.startswith(tuple(skip_prefixes)) or k.endswith("relative_position_index"):
            continue
        out[k] = synth_tensor(k, shp, salt)
    return out


def load_synth_weights(module: torch.nn.Module, salt: int = 0):
    """Overwrites every float parameter/buffer of `module` (except text_encoder.*) in place."""
    sd = module.state_dict()
    new = synth_state_dict({k: v.shape for k, v in sd.items() if v.is_floating_point()}, salt)
    with torch.no_grad():
        for k, v in new.items():
            sd[k].copy_(v.to(sd[k].dtype))
    return module
