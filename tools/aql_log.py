#!/usr/bin/env python3
"""Dumps the AQL dispatch packets the HIP runtime emits for ONE replay of the clip's captured graph (ROCclr's own log:
AMD_LOG_LEVEL=4, AMD_LOG_MASK = LOG_AQL): the packet header carries the barrier bit and the acquire / release fence scopes,
i.e. what orders and publishes memory between graph nodes on the same and on different hardware queues.

    python tools/aql_log.py OUT_DIR            (spawns the logged child before anything touches the GPU)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, argparse, torch
sys.path.insert(0, sys.argv[1])
from tce_rvos_amd import build_model
m, _, _ = build_model(argparse.Namespace(backbone="swin_t_p4w7", with_box_refine=True, binary=True, freeze_text_encoder=True,
                                         f_token=8, qtrans=True, num_feature_levels=4, text_encoder_layers=1))
m = m.cuda().eval()
T, H, W = 2, 96, 128
g = torch.Generator().manual_seed(0)
frames = torch.randn(T, 3, H, W, generator=g).cuda()
ids = torch.randint(3, 50000, (1, 8), generator=g).cuda()
tgt = [{"size": torch.tensor([H, W])}]
for _ in range(3):  # eager, capture + first replay, replay
    m([frames], ids, tgt)
torch.cuda.synchronize()
import glob
log = max(glob.glob(os.environ["AMD_LOG_LEVEL_FILE"] + "*"), key=os.path.getsize)  # ROCclr appends _<pid>
mark = os.path.getsize(log)
m([frames], ids, tgt)   # the replay whose packets are wanted
torch.cuda.synchronize()
print("REPLAY_LOG_OFFSET", mark, os.path.getsize(log), flush=True)
"""


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    log = os.path.join(out, "aql_raw.log")
    if os.path.exists(log):
        os.remove(log)
    env = dict(os.environ, AMD_LOG_LEVEL="4", AMD_LOG_MASK=str(8 | 8192), AMD_LOG_LEVEL_FILE=log)
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=900)
    print(p.stdout[-2000:])
    if p.returncode:
        print(p.stderr[-3000:])
        raise SystemExit(p.returncode)
    off = [l for l in p.stdout.splitlines() if l.startswith("REPLAY_LOG_OFFSET")]
    lo, hi = (int(v) for v in off[-1].split()[1:3])
    # ROCclr may append the pid to the file name
    cands = [os.path.join(out, f) for f in os.listdir(out) if f.startswith("aql_raw.log")]
    print("log files:", [(c, os.path.getsize(c)) for c in cands])
    src = max(cands, key=os.path.getsize)
    with open(src, "rb") as f:
        f.seek(max(lo, 0))
        data = f.read((hi - lo) if hi > lo >= 0 else (8 << 20))
    with open(os.path.join(out, "aql_replay.log"), "wb") as f:
        f.write(data[:24 << 20])
    for c in cands:  # the full log can be hundreds of MB: keep the replay's slice only
        os.remove(c)
    print("replay slice:", len(data), "bytes")
    print(summarise(data.decode(errors="replace")))


def summarise(text):
    """Histogram of the packets of the slice: kind, barrier bit, acquire / release fence scope (0 none, 1 agent, 2 system),
    per hardware queue, and the run-length sequence (q<queue><D|B><acquire><release>x<count>)."""
    import collections
    import re
    pat = re.compile(r"HWq=(0x[0-9a-f]+), id=\d+, (\w+) Header = 0x[0-9a-f]+ \(type=\d+, barrier=(\d), acquire=(\d), release=(\d)\)")
    rows = pat.findall(text)
    qs = {}
    out = [f"{len(rows)} AQL packets"]
    hist = collections.Counter((k, b, a, r) for _, k, b, a, r in rows)
    for (k, b, a, r), n in sorted(hist.items(), key=lambda kv: -kv[1]):
        out.append(f"  {k:10s} barrier={b} acquire={a} release={r}: {n}")
    perq = collections.Counter((qs.setdefault(q, len(qs)), k) for q, k, *_ in rows)
    out.append("  per hardware queue: " + ", ".join(f"q{q} {k} {n}" for (q, k), n in sorted(perq.items())))
    seq, prev, n = [], None, 0
    for q, k, b, a, r in rows:
        cur = f"q{qs[q]}{k[0]}{a}{r}"
        if cur == prev:
            n += 1
        else:
            if prev:
                seq.append(f"{prev}x{n}")
            prev, n = cur, 1
    if prev:
        seq.append(f"{prev}x{n}")
    out.append("  sequence: " + " ".join(seq))
    return "\n".join(out)


if __name__ == "__main__":
    main()
