"""tce_rvos_amd: MI355X-native TCE-RVOS per-clip forward behind the reference's build_model()/forward()
boundary (see DESIGN.md / INTEGRATION.md).  Importing the package does not load the HIP library; the first
kernel call does, and raises if it is missing (there is no fallback path)."""
from .config import ModelConfig, config_from_args, param_shapes  # noqa: F401
from .model import NestedTensor, ReferFormer, build_model, nested_tensor_from_videos_list  # noqa: F401
from .weights import load_synth_weights, synth_state_dict  # noqa: F401
