"""Every distinct GEMM launch of one BASELINE-config-2 clip, timed under each tile variant (graph-timed, so launch
overhead is excluded): shows where the automatic tile choice (select_tile_ex) leaves time on the table."""
import sys, os, argparse, ctypes, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import build_model, ops
from tce_rvos_amd._lib import lib, check, GemmArgs

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="swin_t_p4w7")
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--height", type=int, default=360)
ap.add_argument("--width", type=int, default=640)
args = ap.parse_args()

model, _, _ = build_model(argparse.Namespace(backbone=args.backbone, with_box_refine=True, binary=True, f_token=8, qtrans=True))
model = model.cuda().eval()
model.use_graph = False
T, H, W = args.frames, args.height, args.width
frames = torch.randn(T, 3, H, W, device="cuda")
ids = torch.randint(3, 50000, (1, 32), device="cuda")
tgt = [{"size": torch.tensor([H, W])}]
model([frames], ids, tgt)

recorded = []
orig = ops._gemm_launch
def rec(g):
    c = GemmArgs()
    ctypes.memmove(ctypes.byref(c), ctypes.byref(g), ctypes.sizeof(GemmArgs))
    recorded.append(c)
    orig(g)
ops._gemm_launch = rec
model([frames], ids, tgt)
torch.cuda.synchronize()
ops._gemm_launch = orig

def key(g):
    return (g.M, g.N, g.K, max(1, g.batch), g.conv, bool(g.A2), g.act, g.res_mode, g.kh, g.stride)
groups = {}
for g in recorded:
    groups.setdefault(key(g), []).append(g)
print(f"{len(recorded)} GEMM launches per clip, {len(groups)} distinct shapes")

def time_tile(g, tile, reps=20):
    lib().tce_gemm_force_tile(tile)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            check(lib().tce_gemm_f32(ctypes.byref(g), s.cuda_stream), "gemm")
        s.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(reps):
                check(lib().tce_gemm_f32(ctypes.byref(g), torch.cuda.current_stream().cuda_stream), "gemm")
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
    lib().tce_gemm_force_tile(0)
    return e0.elapsed_time(e1) * 1e-3 / (2 * reps)

TILES = [256128, 128128, 12864, 6464, 6465]
tot_auto = tot_best = 0.0
rows = []
for k, gs in groups.items():
    g = gs[0]
    auto = lib().tce_gemm_select_tile_ex(g.M, g.N, g.K, max(1, g.batch), g.conv)
    t_auto = time_tile(g, 0)
    res = {}
    for t in TILES:
        if t == 256128 and g.conv:
            continue
        res[t] = time_tile(g, t)
    best = min(res, key=res.get)
    n = len(gs)
    tot_auto += n * t_auto
    tot_best += n * min(t_auto, res[best])
    rows.append((n * (t_auto - min(t_auto, res[best])), k, n, auto, t_auto, best, res))
rows.sort(key=lambda r: -r[0])
for gain, k, n, auto, t_auto, best, res in rows:
    M, N, K, b, conv, a2, act, rm, kh, st = k
    fl = 2.0 * M * N * K * b
    print(f"M={M:6d} N={N:5d} K={K:5d} b={b} conv={conv}(k{kh}s{st}) a2={int(a2)} x{n:3d}  auto {auto:6d} {t_auto*1e6:7.1f}us {fl/t_auto/1e12:6.1f}TF | best {best:6d} {res[best]*1e6:7.1f}us | gain/clip {gain*1e6:7.1f}us | "
          + " ".join(f"{t}:{v*1e6:.1f}" for t, v in res.items()), flush=True)
print(f"sum per clip: auto {tot_auto*1e3:.3f} ms, best-per-shape {tot_best*1e3:.3f} ms")
