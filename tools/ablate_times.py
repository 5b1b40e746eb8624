"""Critical-path shares by ablation: runs tools/replay_latency.py with one stage skipped at a time (TCE_ABLATE, results are
garbage then) and prints the steady-state time per clip that disappears.  Each run is a child process (the switch is read at
import)."""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
extra = sys.argv[1:]
cases = ["", "text", "swin2,swin3", "text,swin2,swin3", "swin0,swin1", "decoder", "ftf_tok", "ffn:encoder.ffn", "enc_msda", "ffn:pixel.ffn"]
base = None
for c in cases:
    env = dict(os.environ, TCE_ABLATE=c)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "replay_latency.py"), "--reps", "60"] + extra, env=env,
                         capture_output=True, text=True).stdout
    m = re.search(r"steady state ([0-9.]+) ms", out)
    ms = float(m.group(1)) if m else float("nan")
    if base is None:
        base = ms
    print(f"ablate {c or '(nothing)':<20s} {ms:7.3f} ms per clip   delta {base - ms:+.3f} ms", flush=True)
