"""Ablation + in-kernel stamps of the fused FFN kernel (csrc/chain.hip).

    python tools/ffn_ablate.py --build      (CPU side: compiles the -DFFN_ABL=n variants into tce-rvos_amd/lib/abl/)
    python tools/ffn_ablate.py              (GPU side: times every variant on one shape, prints the stamp breakdown)
"""
import sys, os, argparse, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABL = os.path.join(ROOT, "tce-rvos_amd", "lib", "abl")
CSRC = os.path.join(ROOT, "tce-rvos_amd", "csrc")
VARIANTS = {0: "full", 1: "no DMA in loop", 2: "no barrier", 3: "no DMA, no barrier", 4: "no MFMA", 5: "no MFMA, no DMA",
            7: "no MFMA, no DMA, no barrier (ds_read only)"}

ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--M", type=int, default=24100)
ap.add_argument("--C", type=int, default=256)
ap.add_argument("--Hd", type=int, default=2048)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()

if a.build:
    os.makedirs(ABL, exist_ok=True)
    procs = []
    stub = os.path.join(ABL, "mode_stub.hip")  # the GEMM mode lives in gemm.hip, which these stand-alone builds leave out
    with open(stub, "w") as fh:
        fh.write("int tce_gemm_single_pass() { return 0; }\n")
    for n in VARIANTS:
        out = os.path.join(ABL, f"libffn_abl{n}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", f"-DFFN_ABL={n}",
               os.path.join(CSRC, "chain.hip"), os.path.join(CSRC, "capi.hip"), stub, "-o", out]
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        assert p.wait() == 0
    print("built", sorted(os.listdir(ABL)))
    sys.exit(0)

import torch
torch.manual_seed(0)
M, Cn, Hd = a.M, a.C, a.Hd
x = torch.randn(M, Cn, device="cuda")
w1 = torch.randn(Hd, Cn, device="cuda") / Cn ** 0.5
b1 = torch.randn(Hd, device="cuda") * 0.1
w2 = torch.randn(Cn, Hd, device="cuda") / Hd ** 0.5
b2 = torch.randn(Cn, device="cuda") * 0.1
g = torch.ones(Cn, device="cuda")
be = torch.zeros(Cn, device="cuda")
out = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
for n, name in VARIANTS.items():
    path = os.path.join(ABL, f"libffn_abl{n}.so")
    if not os.path.exists(path):
        continue
    l = C.CDLL(path)
    l.tce_ffn_packed_bytes.restype = i64
    l.tce_ffn_packed_bytes.argtypes = [i32, i32]
    l.tce_ffn_pack_f32.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    l.tce_ffn_fused_f32.argtypes = [vp, i64, vp, vp, vp, vp, f32, vp, vp, f32, vp, i64, i32, i32, i32, i32, vp]
    l.tce_debug_ffn_set_stamp_buffer.argtypes = [vp]
    pk = torch.empty(l.tce_ffn_packed_bytes(Cn, Hd), dtype=torch.uint8, device="cuda")
    assert l.tce_ffn_pack_f32(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), pk.data_ptr(), Cn, Hd, stream) == 0

    def run():
        assert l.tce_ffn_fused_f32(x.data_ptr(), Cn, pk.data_ptr(), b2.data_ptr(), None, None, 1e-5, g.data_ptr(),
                                   be.data_ptr(), 1e-5, out.data_ptr(), Cn, M, Cn, Hd, 1, stream) == 0
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
    st = torch.zeros(1024 * 8, dtype=torch.int64, device="cuda")
    l.tce_debug_ffn_set_stamp_buffer(st.data_ptr())
    run()
    torch.cuda.synchronize()
    l.tce_debug_ffn_set_stamp_buffer(None)
    nwg = min(1024, (M + 127) // 128)
    s = st.view(1024, 8)[:nwg].double().cpu()
    pro, loop, epi = (s[:, 1] - s[:, 0]).median(), (s[:, 2] - s[:, 1]).median(), (s[:, 3] - s[:, 2]).median()
    tot = (s[:, 3] - s[:, 0])
    clk = (tot / ((s[:, 5] - s[:, 4]) * 10e-9)).median() / 1e9  # s_memrealtime ticks at 100 MHz
    print(f"abl {n} {name:44s} {us:8.1f} us | cycles: prologue {pro:7.0f} loop {loop:8.0f} ({loop / (Hd // 32 + 1):6.0f}/iter) "
          f"epilogue {epi:6.0f} | clock {clk:.2f} GHz", flush=True)
