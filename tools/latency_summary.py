"""How latency-bound a clip is, from a rocprofv3 --kernel-trace rocpd database of `python bench.py ...`:
launches per clip, launches shorter than 15 us and what they sum to, share of the time with exactly one kernel in flight.
Clips are counted by the replay's segment-copy launches (one stages the inputs, one hands the outputs back).

    python tools/latency_summary.py DB "BASELINE config 2" [profiles/r04_latency_bound.json]     (merges into the JSON)
"""
import json
import os
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cfg = sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else None
rows = db.execute("select start, end, name from kernels order by start").fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * 0.5  # steady state: skip warm-up, eager and capture passes
rows = [r for r in rows if r[0] >= lo]
# cut at clip boundaries: from the first input-staging copy in the window to the last one
cp = [i for i, r in enumerate(rows) if "copy_segments" in r[2]]
stage = cp[0::2] if len(cp) >= 4 else []
if len(stage) >= 3:
    rows = rows[stage[0]:stage[-1]]
    clips = len(stage) - 1
else:
    clips = None
wall = max(r[1] for r in rows) - rows[0][0]
ev = sorted([(r[0], 1) for r in rows] + [(r[1], -1) for r in rows])
busy, depth, last = {}, 0, ev[0][0]
for t, d in ev:
    busy[depth] = busy.get(depth, 0) + (t - last)
    depth, last = depth + d, t
tot = float(sum(busy.values()))
short = [r for r in rows if r[1] - r[0] < 15000]
ksum = sum(r[1] - r[0] for r in rows)
n = float(clips or 1)
res = {"clips_in_window": clips, "launches_per_clip": round(len(rows) / n, 1),
       "launches_shorter_than_15us_per_clip": round(len(short) / n, 1),
       "their_summed_kernel_ms_per_clip": round(sum(r[1] - r[0] for r in short) / n / 1e6, 3),
       "kernel_busy_ms_per_clip": round(ksum / n / 1e6, 3), "wall_ms_per_clip_under_profiler": round(wall / n / 1e6, 3),
       "share_of_time_one_kernel_in_flight": round(busy.get(1, 0) / tot, 3),
       "share_of_time_idle": round(busy.get(0, 0) / tot, 3), "average_kernels_in_flight": round(ksum / wall, 2)}
print(cfg, json.dumps(res))
if out:
    d = {}
    if os.path.exists(out):
        with open(out) as f:
            d = json.load(f)
    d[cfg] = res
    with open(out, "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
