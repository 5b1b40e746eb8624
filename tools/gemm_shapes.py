"""Every GEMM launch of one eager clip (config 2 by default) with its shape, pipeline stage and duration, grouped."""
import sys, os, argparse, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import build_model, pipeline, ops

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="swin_t_p4w7")
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--height", type=int, default=360)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--max-rows", type=int, default=4000, help="list GEMMs with at most this many rows")
ap.add_argument("--group", type=int, default=1, help="clips per forward (forward_group)")
a = ap.parse_args()
model, _, _ = build_model(argparse.Namespace(backbone=a.backbone, with_box_refine=True, binary=True, f_token=8, qtrans=True))
model = model.cuda().eval()
model.use_graph = False
frames = torch.randn(a.frames, 3, a.height, a.width, device="cuda")
ids = torch.randint(3, 50000, (a.group, 32), device="cuda")
tgt = [{"size": torch.tensor([a.height, a.width])}]
clips = [frames] + [torch.randn_like(frames) for _ in range(a.group - 1)]


def run():
    return model([frames], ids, tgt) if a.group == 1 else model.forward_group(clips, ids, tgt)


for _ in range(2):
    run()
log = []
orig = ops._gemm_launch


def logged(g, splitk=1, ws=None, *rest, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(g, splitk, ws, *rest, **kw)
    e1.record()
    stage = pipeline.STAGE_EVENTS[-1][0] if pipeline.STAGE_EVENTS else "?"
    log.append((stage, g.M, g.N, g.K, max(1, g.batch), int(g.conv), splitk, torch.cuda.current_stream().cuda_stream, e0, e1))


ops._gemm_launch = logged
pipeline.STAGE_EVENTS = []
run()
torch.cuda.synchronize()
main = torch.cuda.current_stream().cuda_stream
agg = collections.OrderedDict()
for stage, M, N, K, b, conv, sk, st, e0, e1 in log:
    key = (stage, M, N, K, b, conv, sk, "main" if st == main else "side")
    v = agg.setdefault(key, [0, 0.0])
    v[0] += 1
    v[1] += e0.elapsed_time(e1) * 1e3
print(f"{len(log)} GEMM launches; those with M <= {a.max_rows}: (stage after which they run, M, N, K, batch, conv, splitk, stream) x count, total us")
tot = 0.0
from tce_rvos_amd._lib import lib
for k, v in agg.items():
    if k[1] <= a.max_rows:
        tot += v[1]
        tile = lib().tce_gemm_select_tile_ex(k[1], k[2], k[3], k[4], k[5])
        tf = 2.0 * k[1] * k[2] * k[3] * k[4] * (9 if k[5] else 1) * v[0] / v[1] / 1e6
        print(f"  {str(k):90s} x{v[0]:3d} {v[1]:8.1f} us  tile {tile}  {tf:6.1f} TFLOP/s")
print(f"small-M total {tot:.0f} us")
