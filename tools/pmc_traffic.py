"""Aggregates two rocprofv3 counter passes into profiles/r03_pmc_traffic.json (PMC_OUT overrides the file name) (HBM-side bytes per kernel launch).

    TCE_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-variants
    TCE_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py  (same flags)
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv [clips]

Counters are KiB per dispatch.  FETCH_SIZE is doubled: on gfx950 it tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  Infinity-Cache hits are counted, so this is traffic on
the L2's memory side, an upper bound on HBM bytes."""
import csv, json, os, sys, collections

def load(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc

fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
clips = int(sys.argv[3]) if len(sys.argv) > 3 else 4
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `TCE_GRAPH=0 python3 bench.py "
               "--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-variants` (4 eager clips); counters are KiB per dispatch; FETCH_SIZE doubled "
               "(gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md section HBM); L2 memory-side traffic, "
               "Infinity-Cache hits included", "kernels": {}}
for name in sorted(set(fetch) | set(write)):
    f, w = fetch.get(name, [0.0, 0]), write.get(name, [0.0, 0])
    n = max(f[1], w[1], 1)
    fb = 2.0 * 1024.0 * f[0] / max(f[1], 1)
    wb = 1024.0 * w[0] / max(w[1], 1)
    key = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:120]
    out["kernels"][key] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                           "hbm_bytes_per_launch": fb + wb}
tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out["kernels"].values())
out["clips"] = clips
out["hbm_bytes_per_clip_all_kernels"] = tot / clips
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", os.environ.get("PMC_OUT", "r03_pmc_traffic.json"))
json.dump(out, open(dst, "w"), indent=1)
print(f"total L2 memory-side traffic: {tot / clips / 1e9:.2f} GB per clip over {clips} clips")
big = sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]
for k, v in big:
    print(f"{k[:90]:90s} x{v['launches']:5d}  fetch {v['fetch_bytes_per_launch']/1e6:9.2f} MB  write {v['write_bytes_per_launch']/1e6:9.2f} MB")
