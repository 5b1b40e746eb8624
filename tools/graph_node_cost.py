"""What a kernel boundary costs inside a replayed hipGraph on this box (DESIGN 3.6 / VERDICT r2 next #4): chains of N tiny
dependent launches (a 40 x 256 copy) captured (a) on one stream, (b) with every second launch forked to a side stream
and joined again (the shape of the frame-token path's parallel branches), (c) two independent chains on two streams.
Prints us per node."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import ops

N = 200
a = torch.randn(40, 256, device="cuda")
bufs = [torch.empty(40, 256, device="cuda") for _ in range(4)]
side = torch.cuda.Stream()


def chain_single():
    src = a
    for i in range(N):
        ops.tile(src, 1, out=bufs[i & 1])
        src = bufs[i & 1]


def chain_forked():
    cur = torch.cuda.current_stream()
    src = a
    for i in range(N // 2):
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            ops.tile(src, 1, out=bufs[2])
        ops.tile(src, 1, out=bufs[i & 1])
        cur.wait_stream(side)
        src = bufs[i & 1]


def chain_two():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        src = a
        for i in range(N // 2):
            ops.tile(src, 1, out=bufs[2 + (i & 1)])
            src = bufs[2 + (i & 1)]
    src = a
    for i in range(N // 2):
        ops.tile(src, 1, out=bufs[i & 1])
        src = bufs[i & 1]
    cur.wait_stream(side)


def time_graph(fn, name):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:<46s} {us:8.1f} us per replay  {us / N:6.2f} us per node ({N} nodes)", flush=True)
    return g


def time_eager(fn, name):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print(f"{name:<46s} {us:8.1f} us per pass    {us / N:6.2f} us per node ({N} nodes)", flush=True)


keep = [time_graph(chain_single, "graph, one stream, dependent chain"),
        time_graph(chain_forked, "graph, fork + join around every second node"),
        time_graph(chain_two, "graph, two independent chains on two streams")]
time_eager(chain_single, "eager, one stream, dependent chain")
