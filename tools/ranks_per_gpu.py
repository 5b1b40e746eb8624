#!/usr/bin/env python3
"""Aggregate clips/s of R processes sharing ONE GPU (VERDICT r3 #3b).

One process's graph replays are fed through that process's four hardware queues in submission order (DESIGN.md section 3.8:
two clips in flight inside one process gain 1-2 %); separate processes own separate queues.  Each of the R ranks runs
bench.py at BASELINE config 2 on cuda:0 (`--ranks-per-gpu R`), gloo for the barrier / max-over-ranks timing, no per-step
gather (`--no-gather`: gloo would add a host copy per step).  Prints one line per R; run from a process that has not
touched the GPU.

    python tools/ranks_per_gpu.py [--ranks 1 2 3] [--steps 250] [-- extra bench.py arguments]
"""
import argparse
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(R, steps, extra):
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(R):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(R), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TCE_BENCH_FORCE_DIST="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(R), "--ranks-per-gpu", str(R),
                                       "--backend", "gloo", "--no-gather", "--steps", str(steps), "--warmup", "10",
                                       "--no-cpu-baseline", "--no-roofline", "--no-variants"] + extra,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1500) for p in procs]
    if any(p.returncode for p in procs):
        raise SystemExit("\n".join(o[1][-2000:] for o in outs))
    return json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--steps", type=int, default=250)
    args, extra = ap.parse_known_args()
    extra = [e for e in extra if e != "--"]
    base = None
    for R in args.ranks:
        line = run(R, args.steps, extra)
        base = base or line["value"]
        print(f"ranks per GPU {R}: aggregate {line['value']:8.2f} clips/s  ({line['ms_per_step']:.3f} ms per step of {R} clips, "
              f"x{line['value'] / base:.3f} of one rank)", flush=True)


if __name__ == "__main__":
    main()
