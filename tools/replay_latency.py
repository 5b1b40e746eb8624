"""Host cost of one graph replay, isolated latency of one clip and steady-state time per clip at config 2 (or --args of
bench.py's shape flags): is a replay bound by the host's packet submission, by its own dependency chain, or by the GPU?
    python tools/replay_latency.py [--backbone ... --frames T --height H --width W]"""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import build_model

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="swin_t_p4w7")
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--height", type=int, default=360)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--reps", type=int, default=100)
a = ap.parse_args()
ns = argparse.Namespace(backbone=a.backbone, with_box_refine=True, binary=True, freeze_text_encoder=True, f_token=8, qtrans=True,
                        num_feature_levels=4)
model, _, _ = build_model(ns)
model = model.cuda().eval()
g = torch.Generator().manual_seed(1)
clip = torch.randn(a.frames, 3, a.height, a.width, generator=g).cuda()
ids = torch.randint(3, 50264, (1, 32), generator=g)
ids[0, 0], ids[0, -1] = 0, 2
ids = ids.cuda()
tgt = [{"size": torch.tensor([a.height, a.width])}]
for _ in range(4):
    model([clip], ids, tgt)
torch.cuda.synchronize()
# steady state: back-to-back replays
t0 = time.perf_counter()
host = 0.0
for _ in range(a.reps):
    h0 = time.perf_counter()
    model([clip], ids, tgt)
    host += time.perf_counter() - h0
torch.cuda.synchronize()
steady = (time.perf_counter() - t0) / a.reps
# isolated: the GPU is idle when the replay is submitted
iso = 0.0
for _ in range(20):
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    model([clip], ids, tgt)
    torch.cuda.synchronize()
    iso += time.perf_counter() - h0
# host cost of the call alone when the GPU is idle (no waiting on a full queue)
hidle = 0.0
for _ in range(20):
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    model([clip], ids, tgt)
    hidle += time.perf_counter() - h0
print(f"steady state {steady * 1e3:.3f} ms per clip ({1 / steady:.1f} clips/s); host time inside forward() {host / a.reps * 1e3:.3f} ms per clip "
      f"(GPU busy) / {hidle / 20 * 1e3:.3f} ms (GPU idle); isolated clip latency {iso / 20 * 1e3:.3f} ms", flush=True)
