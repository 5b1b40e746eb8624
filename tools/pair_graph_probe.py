"""Do n clips run side by side when their captured graphs are the branches of ONE executable (tce_graph_group)?
Captures the config-2 clip in n slots (own arenas each), builds the group executable, checks its outputs against the
single replays bit for bit and times single replays back to back against group launches.
    TCE_KEEP_GRAPHS=1 python tools/pair_graph_probe.py [--n 2] [--reps 60]"""
import argparse
import ctypes as C
import os
import sys
import time
os.environ["TCE_KEEP_GRAPHS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import build_model, ops
from tce_rvos_amd._lib import check, lib
from tce_rvos_amd.model import _flat_outputs

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="swin_t_p4w7")
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--height", type=int, default=360)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--n", type=int, default=2)
ap.add_argument("--reps", type=int, default=60)
a = ap.parse_args()
ns = argparse.Namespace(backbone=a.backbone, with_box_refine=True, binary=True, freeze_text_encoder=True, f_token=8, qtrans=True,
                        num_feature_levels=4)
model, _, _ = build_model(ns)
model = model.cuda().eval()
g = torch.Generator().manual_seed(1)
clips = [torch.randn(a.frames, 3, a.height, a.width, generator=g).cuda() for _ in range(a.n)]
ids = torch.randint(3, 50264, (1, 32), generator=g)
ids[0, 0], ids[0, -1] = 0, 2
ids = ids.cuda()
tgt = [{"size": torch.tensor([a.height, a.width])}]
single = []
for s_ in range(a.n):
    for _ in range(4):
        out = model([clips[s_]], ids, tgt, slot=s_)
    single.append([t.clone() for t in _flat_outputs(out)])
torch.cuda.synchronize()
ents = [e for k, e in model._graphs.items() if k[0] == "clip"]
assert len(ents) == a.n, (len(ents), list(model._graphs))
ents.sort(key=lambda e: 0)  # insertion order = slot order
raw = (C.c_void_p * a.n)(*[int(e[0].raw_cuda_graph()) for e in ents])
ex = C.c_void_p()
t0 = time.perf_counter()
check(lib().tce_graph_group(raw, a.n, C.byref(ex)), "tce_graph_group")
print(f"group executable of {a.n} clips built in {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
stream = torch.cuda.current_stream()


def launch_group():
    for e, c in zip(ents, clips):
        ops.copy_many(e[1], (c, ids))
    check(lib().tce_graph_launch(ex, stream.cuda_stream), "tce_graph_launch")


launch_group()
torch.cuda.synchronize()
worst = 0.0
for e, ref in zip(ents, single):
    for t, r in zip(_flat_outputs(e[2]), ref):
        worst = max(worst, float((t - r).abs().max()))
print(f"group launch vs single replays: max |d| over all outputs {worst:.3e}", flush=True)


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def singles():
    for s_ in range(a.n):
        model([clips[s_]], ids, tgt, slot=s_)


ts = timed(singles, a.reps) / a.n
tg = timed(launch_group, a.reps) / a.n
print(f"single replays back to back {ts * 1e3:.3f} ms per clip ({1 / ts:.1f} clips/s);  group of {a.n}: {tg * 1e3:.3f} ms per clip "
      f"({1 / tg:.1f} clips/s, x{ts / tg:.3f})", flush=True)
