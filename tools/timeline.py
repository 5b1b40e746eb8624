"""One steady-state clip of a rocprofv3 --kernel-trace run as a timeline: every kernel between two consecutive input-staging
launches (copy_segments_kernel with the smaller grid = the replay's input copy) with its start offset, duration and queue.
    python tools/timeline.py gpurun_out/x/prof/x_results.db [which_clip_from_the_end=3] [min_us=0]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows = db.execute("select start, end, queue_id, name, grid_x from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "copy_segments" in r[3]]
# a replay = [input copy] graph [output copy]: two copy launches per clip; take pairs
starts = marks[0::2] if len(marks) % 2 == 0 else marks[1::2]
i0 = starts[-back - 1]
i1 = starts[-back]
clip = rows[i0:i1]
t0 = clip[0][0]
print(f"clip of {len(clip)} kernels, {(clip[-1][1] - t0) / 1e3:.1f} us from first start to last end; next clip starts at {(rows[i1][0] - t0) / 1e3:.1f} us")
last_end = t0
for s, e, q, n, gx in clip:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n)[:60]
    if (e - s) / 1e3 >= min_us:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  q{q}  {n}")
