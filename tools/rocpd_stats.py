"""Per-kernel summary (calls, total, average) from a rocprofv3 rocpd sqlite database -> CSV on stdout.
    python tools/rocpd_stats.py gpurun_out/x/prof/x_results.db [clips]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
clips = float(sys.argv[2]) if len(sys.argv) > 2 else None
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"' + (',"UsPerClip"' if clips else ""))
for n, c, t, mn, mx in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    line = f'"{n[:160]}",{c},{t},{t / c:.1f},{100.0 * t / tot:.2f},{mn},{mx}'
    if clips:
        line += f",{t / clips / 1e3:.1f}"
    print(line)
