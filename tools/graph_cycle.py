"""Cycles N distinct clip shapes through a model whose graph cache holds 3 entries: every shape runs eagerly, is captured, and is
replayed (bit-identical to the eager pass), then is evicted -- with OWNED executables (model.GRAPH_OWN_EXEC: nodes re-created in a
fresh graph, tce_graph_group) the evicted executable is DESTROYED, nothing accumulates and the capture budget is never touched.
Run in a process of its own (tests/test_e2e_gpu.py::test_graph_executables_are_destroyed_on_eviction): a runtime crash here must
not take a test session down.
    python tools/graph_cycle.py [--shapes 600] [--revisit 20]"""
import argparse
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import build_model
from tce_rvos_amd import model as M

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", type=int, default=600)
ap.add_argument("--revisit", type=int, default=20, help="every this many shapes, go back to the first shape (re-capture after eviction)")
a = ap.parse_args()
ns = argparse.Namespace(backbone="swin_t_p4w7", with_box_refine=True, binary=True, freeze_text_encoder=True, f_token=8, qtrans=True,
                        num_feature_levels=4, text_encoder_layers=1)
model, _, _ = build_model(ns)
model = model.cuda().eval()
model.max_graphs = 3
ids = torch.arange(3, 10)[None].cuda()
shapes = [(1 + i % 2, 64 + 4 * ((i // 2) % 20), 64 + 4 * (i // 40)) for i in range(a.shapes)]
assert len(set(shapes)) == a.shapes
g = torch.Generator().manual_seed(0)
base = torch.randn(2, 3, 64 + 4 * 20, 64 + 4 * (a.shapes // 40 + 1), generator=g).cuda()
first_out = None
t0 = time.time()
worst_cached = 0
for i, (T, H, W) in enumerate(shapes):
    clip = base[:T, :, :H, :W].contiguous()
    tgt = [{"size": torch.tensor([H, W])}]
    outs = [model([clip], ids, tgt)["pred_masks"].clone() for _ in range(3)]   # eager, capture (+ its replay), replay
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), ("replay != eager", i, T, H, W)
    if i == 0:
        first_out = outs[0]
    if a.revisit and i and i % a.revisit == 0:   # the first shape was evicted long ago: eager (sighting known) -> re-captured
        T0, H0, W0 = shapes[0]
        c0 = base[:T0, :, :H0, :W0].contiguous()
        for _ in range(2):
            assert torch.equal(model([c0], ids, [{"size": torch.tensor([H0, W0])}])["pred_masks"], first_out), ("revisit", i)
    worst_cached = max(worst_cached, len(model._graphs))
    if i % 100 == 0:
        print(f"# {i} shapes, {time.time() - t0:.0f} s, {M.graph_state()}", flush=True)
st = M.graph_state()
print(json.dumps({"shapes": a.shapes, "seconds": round(time.time() - t0, 1), "max_cached_entries": worst_cached, "graph_state": st,
                  "own_exec": M.GRAPH_OWN_EXEC}))
