"""Time of the RoBERTa branch alone (12 layers, 32 tokens), eager launches on an idle GPU: TCE_THIN=0 (tiled split-K GEMMs) vs
TCE_THIN=1 (weight-stream launches, csrc/thin.hip).   python tools/text_bench.py"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tce_rvos_amd import build_model, ops  # noqa: E402

model, _, _ = build_model(argparse.Namespace(backbone="swin_t_p4w7", with_box_refine=True, binary=True, f_token=8, qtrans=True))
model = model.cuda().eval()
model._ensure_packed()
ids = torch.randint(3, 50000, (1, 32), device="cuda")
ar = ops.Arena("cuda", 64 << 20)
plan = model._text_plan()


def run():
    ar.reset()
    with model.arith("text"):
        return plan.forward(ids, ar.alloc)


for thin in (False, True, False, True):
    ops.THIN_ENABLED = thin
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"TCE_THIN={int(thin)}: text branch {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per forward (graph replay, idle GPU)", flush=True)
