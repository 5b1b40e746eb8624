"""Shared helpers for the tests (fixtures -> oracle inputs)."""
import json
import os

import numpy as np
import torch

from tce_rvos_amd.weights import synth_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def manifest(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def synth_sd_from_manifest(name, salt):
    man = manifest(name)
    return synth_state_dict({k: v[0] for k, v in man.items() if v[1].startswith("float")}, salt)


def synth_frames(T, H, W, seed):
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(int(T), 3, int(H), int(W), generator=g)
