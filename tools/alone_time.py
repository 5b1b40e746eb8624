"""Which kernels run ALONE?  Over the steady-state clips of a rocprofv3 --kernel-trace run (bench.py, B=1): per kernel name, the
time during which it was the only kernel in flight, the time it shared the GPU, and the idle time in front of it.
    python tools/alone_time.py x_results.db [clips_from_the_end=40]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
nclips = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = db.execute("select start, end, name, grid_x from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "copy_segments" in r[2]]
starts = marks[0::2] if len(marks) % 2 == 0 else marks[1::2]
i0, i1 = starts[-nclips - 1], starts[-1]
clip = rows[i0:i1]
T0, T1 = clip[0][0], rows[i1][0]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:70]


ev = []
for k, (s, e, n, gx) in enumerate(clip):
    ev.append((s, 1, k))
    ev.append((e, 0, k))
ev.sort()
alone = collections.Counter()
shared = collections.Counter()
gap_before = collections.Counter()
calls = collections.Counter()
live = set()
prev = T0
idle_since = None
for t, kind, k in ev:
    dt = t - prev
    if dt > 0:
        if len(live) == 1:
            alone[short(clip[next(iter(live))][2])] += dt
        elif len(live) > 1:
            for j in live:
                shared[short(clip[j][2])] += dt
    prev = t
    if kind == 1:
        if not live and idle_since is not None:
            gap_before[short(clip[k][2])] += t - idle_since
        live.add(k)
        calls[short(clip[k][2])] += 1
    else:
        live.discard(k)
        if not live:
            idle_since = t
wall = (T1 - T0) / nclips / 1e3
print(f"{nclips} clips, {wall:.1f} us per clip under the profiler; per clip: alone {sum(alone.values()) / nclips / 1e3:.0f} us, "
      f"idle {sum(gap_before.values()) / nclips / 1e3:.0f} us")
print(f"{'kernel':70s} {'calls':>6s} {'alone us':>9s} {'shared us':>9s} {'idle before us':>14s}   (per clip)")
for n, v in sorted(alone.items(), key=lambda kv: -(kv[1] + gap_before[kv[0]])):
    print(f"{n:70s} {calls[n] / nclips:6.1f} {v / nclips / 1e3:9.1f} {shared[n] / nclips / 1e3:9.1f} {gap_before[n] / nclips / 1e3:14.1f}")
