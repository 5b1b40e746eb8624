"""Concurrency summary of a rocprofv3 --kernel-trace run (rocpd sqlite database): how many kernels are in flight over the
steady-state part of the run, on how many hardware queues, and how much of the wall time the GPU is idle.
    python tools/overlap_stats.py gpurun_out/x/prof/x_results.db [skip_fraction=0.5]
Used for VERDICT r2 next #3d (why clips-in-flight > 1 does or does not raise throughput)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = db.execute("select start, end, queue_id, stream_id, name from kernels order by start").fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * skip          # skip warm-up / capture passes
rows = [r for r in rows if r[0] >= lo]
wall = max(r[1] for r in rows) - rows[0][0]
ev = []
for s, e, q, st, n in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
busy = {}
depth, last = 0, ev[0][0]
for t, d in ev:
    busy[depth] = busy.get(depth, 0) + (t - last)
    depth += d
    last = t
tot = sum(busy.values())
ksum = sum(r[1] - r[0] for r in rows)
print(f"kernels {len(rows)}  wall {wall / 1e6:.2f} ms  sum of kernel durations {ksum / 1e6:.2f} ms  average depth {ksum / wall:.2f}")
print("time share by number of kernels in flight: " + "  ".join(f"{k}: {100.0 * v / tot:.1f}%" for k, v in sorted(busy.items()) if v / tot > 0.002))
queues = {}
for s, e, q, st, n in rows:
    a = queues.setdefault(q, [0, 0])
    a[0] += 1
    a[1] += e - s
print("hardware queues used: " + "  ".join(f"q{q}: {c} kernels, {100.0 * d / wall:.0f}% busy" for q, (c, d) in sorted(queues.items())))
print(f"distinct streams: {len(set(r[3] for r in rows))}")
# gaps: time with nothing in flight, split by length
gaps = []
depth, last = 0, None
for t, d in ev:
    if depth == 0 and last is not None and t > last:
        gaps.append(t - last)
    depth += d
    if depth == 0:
        last = t
print(f"idle (depth 0): {100.0 * busy.get(0, 0) / tot:.1f}% of wall in {len(gaps)} gaps; "
      f"gaps > 5 us: {sum(1 for g in gaps if g > 5000)} totalling {sum(g for g in gaps if g > 5000) / 1e6:.2f} ms")
