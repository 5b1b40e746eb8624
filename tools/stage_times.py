"""Main-stream time per pipeline stage of one clip (eager launches, side branches still overlap on their stream)."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import build_model, pipeline

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="swin_t_p4w7")
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--height", type=int, default=360)
ap.add_argument("--width", type=int, default=640)
a = ap.parse_args()
model, _, _ = build_model(argparse.Namespace(backbone=a.backbone, with_box_refine=True, binary=True, f_token=8, qtrans=True))
model = model.cuda().eval()
model.use_graph = False
frames = torch.randn(a.frames, 3, a.height, a.width, device="cuda")
ids = torch.randint(3, 50000, (1, 32), device="cuda")
tgt = [{"size": torch.tensor([a.height, a.width])}]
for _ in range(3):
    model([frames], ids, tgt)
acc = {}
N = 10
for _ in range(N):
    pipeline.STAGE_EVENTS = []
    model([frames], ids, tgt)
    torch.cuda.synchronize()
    ev = pipeline.STAGE_EVENTS
    for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
pipeline.STAGE_EVENTS = None
tot = sum(acc.values())
for k, v in acc.items():
    print(f"{k:28s} {v / N:7.3f} ms  {100 * v / tot:5.1f} %")
print(f"{'total (eager, main stream)':28s} {tot / N:7.3f} ms")
