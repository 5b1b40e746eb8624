"""Soak: N graph replays of one config-2 clip, every output compared bit for bit with the first replay and with the eager pass.
    python tools/replay_soak.py [--reps 300] [--backbone ...] [--group G]     (--group: forward_group of G clips per forward)"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import build_model
from tce_rvos_amd.model import _flat_outputs

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="swin_t_p4w7")
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--height", type=int, default=360)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--group", type=int, default=1)
a = ap.parse_args()
ns = argparse.Namespace(backbone=a.backbone, with_box_refine=True, binary=True, freeze_text_encoder=True, f_token=8, qtrans=True,
                        num_feature_levels=4)
model, _, _ = build_model(ns)
model = model.cuda().eval()
g = torch.Generator().manual_seed(1)
G = max(1, a.group)
clips = [[torch.randn(a.frames, 3, a.height, a.width, generator=g).cuda() for _ in range(G)] for _ in range(2)]
ids = torch.randint(3, 50264, (G, 32), generator=g)
ids[:, 0], ids[:, -1] = 0, 2
ids = ids.cuda()
tgt = [{"size": torch.tensor([a.height, a.width])}]


def run(c):
    if G == 1:
        return _flat_outputs(model([c[0]], ids, tgt))
    return [t for o in model.forward_group(c, ids, tgt) for t in _flat_outputs(o)]


model.use_graph = False
eager = [[t.clone() for t in run(c)] for c in clips]
model.use_graph = True
for _ in range(4):
    for c in clips:
        run(c)
torch.cuda.synchronize()
st_ = model.graph_state()
assert st_["captured"] + st_["owned_alive"] >= 1, st_
bad = 0
for r in range(a.reps):
    for c, e in zip(clips, eager):  # alternate two clips so that a replay never sees its own previous input
        out = run(c)
        if not all(torch.equal(x, y) for x, y in zip(out, e)):
            bad += 1
            worst = max(float((x - y).abs().max()) for x, y in zip(out, e))
            print(f"replay {r}: differs from the eager pass, max |d| {worst:.3e}", flush=True)
print(f"{2 * a.reps} replays ({a.backbone} T={a.frames} {a.height}x{a.width}, {G} clip(s) per forward): {bad} differ from the eager pass", flush=True)
sys.exit(1 if bad else 0)
