"""Ablation of the pixel-stationary 3x3 convolution kernel (csrc/chain.hip).

    python tools/conv3_ablate.py --build    (CPU side: compiles the -DCONV_ABL=n variants into tce-rvos_amd/lib/abl/)
    python tools/conv3_ablate.py            (GPU side: times every variant)
"""
import sys, os, argparse, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABL = os.path.join(ROOT, "tce-rvos_amd", "lib", "abl")
CSRC = os.path.join(ROOT, "tce-rvos_amd", "csrc")
VARIANTS = {0: "full", 2: "no barrier", 4: "no MFMA", 8: "no weight loads in loop",
            16: "no operand loads in loop", 24: "no weight loads, no operand loads", 26: "no loads, no barrier (MFMA + ds_read/ds_write)",
            30: "ds_read + split only"}

ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--T", type=int, default=5)
ap.add_argument("--H", type=int, default=90)
ap.add_argument("--W", type=int, default=160)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()

if a.build:
    os.makedirs(ABL, exist_ok=True)
    procs = []
    stub = os.path.join(ABL, "mode_stub.hip")  # the GEMM mode lives in gemm.hip, which these stand-alone builds leave out
    with open(stub, "w") as fh:
        fh.write("int tce_gemm_single_pass() { return 0; }\n")
    for n in VARIANTS:
        out = os.path.join(ABL, f"libconv_abl{n}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", f"-DCONV_ABL={n}",
               os.path.join(CSRC, "chain.hip"), os.path.join(CSRC, "capi.hip"), stub, "-o", out]
        procs.append(subprocess.Popen(cmd))
    for n, p in zip(VARIANTS, procs):
        assert p.wait() == 0, f"variant {n} failed to build"
    print("built", sorted(f for f in os.listdir(ABL) if f.startswith("libconv")))
    sys.exit(0)

import torch
torch.manual_seed(0)
T, H, W = a.T, a.H, a.W
M = T * H * W
x = torch.randn(M, 256, device="cuda")
w = torch.randn(256, 2304, device="cuda") / 48.0
out = torch.empty(M, 256, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
for n, name in VARIANTS.items():
    path = os.path.join(ABL, f"libconv_abl{n}.so")
    if not os.path.exists(path):
        continue
    l = C.CDLL(path)
    l.tce_conv3x3_packed_bytes.restype = i64
    l.tce_conv3x3_packed_bytes.argtypes = [i32, i32]
    l.tce_conv3x3_pack_f32.argtypes = [vp, vp, i32, i32, vp]
    l.tce_conv3x3_f32.argtypes = [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]
    pk = torch.empty(l.tce_conv3x3_packed_bytes(256, 256), dtype=torch.uint8, device="cuda")
    assert l.tce_conv3x3_pack_f32(w.data_ptr(), pk.data_ptr(), 256, 256, stream) == 0

    def run():
        assert l.tce_conv3x3_f32(x.data_ptr(), 256, pk.data_ptr(), None, out.data_ptr(), 256, T, H, W, 256, 256, stream) == 0
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    print(f"abl {n:2d} {name:48s} {us:8.1f} us   {3 * 2.0 * M * 256 * 2304 / us / 1e6:7.1f} TFLOP/s issued-equivalent", flush=True)
