"""MFMA utilisation and the clock a kernel holds, per kernel, from ONE rocprofv3 counter pass over an eager clip:

    TCE_GRAPH=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d D -o m -- \
        python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-variants
    python tools/pmc_mfma.py D/.../m_counter_collection.csv [out.json]

SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs (MI355X_MICROARCH.md:
"= 32 x N_mfma for 32x32x16"); GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the dispatch was active, so the effective
clock is GRBM_GUI_ACTIVE / 8 / wall time and
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)          (share of all SIMD-cycles with the matrix pipe busy)
(reads high on dispatches shorter than ~0.3 ms, see the guide's DVFS note: the clock column is indicative there)."""
import collections
import csv
import json
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
dur = collections.defaultdict(float)
seen = set()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
        name = re.sub(r"\(.*", "", name)[:110]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], r["Kernel_Name"])
        if key not in seen:
            seen.add(key)
            cnt[name] += 1
            dur[name] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for name, c in acc.items():
    n = cnt[name]
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if not gui:
        continue
    out[name] = {"launches": n, "avg_us": dur[name] / n / 1e3, "total_us": dur[name] / 1e3,
                 "mfma_busy_cycles_per_launch": mf / n, "gui_active_per_launch": gui / n,
                 "clock_ghz": gui / 8.0 / dur[name] if dur[name] else None,
                 "mfma_util": mf / (1024.0 * gui / 8.0)}
rows = sorted(out.items(), key=lambda kv: -kv[1]["total_us"])
print(f"{'kernel':92s} {'n':>5s} {'avg us':>9s} {'MFMA util':>9s} {'clock GHz':>9s}")
for k, v in rows[:28]:
    print(f"{k[:92]:92s} {v['launches']:5d} {v['avg_us']:9.1f} {v['mfma_util']:9.3f} {v['clock_ghz']:9.2f}")
if len(sys.argv) > 2:
    json.dump({"note": __doc__.split("\n\n")[1], "kernels": dict(rows)}, open(sys.argv[2], "w"), indent=1)
