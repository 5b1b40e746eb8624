#!/usr/bin/env python3
"""Throughput benchmark of the north-star path: per-clip forward of TCE-RVOS on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one B=1 clip forward on every rank: Swin-T backbone, T=5 frames of 360x640, a 32-token text
(RoBERTa-base, random init, evaluated every step as the reference does), FTF/IQT deformable transformer,
cross-modal FPN, dynamic mask head -- BASELINE.json config 2 -- followed (N > 1) by the RCCL all-gather of the
per-clip mask logits.  Clips are independent units sharded over ranks (weak scaling, no data-path collective
besides the gather).  Inputs and weights are synthetic and already resident in HBM when the timed region starts.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     : the dominant kernel (fp32 MFMA GEMM, 128x128 tile) -- algorithmic FLOPs of its launches in one
                 step / their summed launch durations, measured with events on the launch stream in a second,
                 instrumented pass over the same K steps (the headline `value` comes from the un-instrumented pass).
  cpu_baseline : the CPU oracle (a port of the reference path) timed on this box's host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "clips/s (T=5, 360×640, Swin-T) at 1/2/4/8 MI355X; mask IoU vs ref"


def _pmc_traffic(prefix):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (separate rocprofv3 --pmc passes of
    this same command: profiles/r01_pmc_traffic.json).  None when no summary matches."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            k = json.load(f)["kernels"]
        hits = [v for name, v in k.items() if name.startswith(prefix) and ", false, false" in name]
        return round(hits[0]["hbm_bytes_per_launch"]) if hits else None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--height", type=int, default=360)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--tokens", type=int, default=32)
    ap.add_argument("--backbone", default="swin_t_p4w7")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--gemm-mode", default="f16x3", choices=["f32", "f16x3"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo (+ TCE_BENCH_ONE_DEVICE=1: every rank on cuda:0) rehearses the N>1 control flow on a 1-GPU box")
    ap.add_argument("--clips-in-flight", type=int, default=1,
                    help="independent B=1 clip forwards kept in flight per GPU and per step (one stream + replay slot each)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    if os.environ.get("TCE_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from tce_rvos_amd import build as _b
    if not os.path.exists(_b.LIB):  # fresh checkout: the shared object is git-ignored; rank 0 builds, the others wait
        if local_rank == 0:
            _b.build(verbose=False)
        if world > 1:
            dist.barrier()
    from tce_rvos_amd import build_model, ops
    from tce_rvos_amd.dist import gather_clip_masks_async

    margs = argparse.Namespace(backbone=args.backbone, with_box_refine=True, binary=True, freeze_text_encoder=True,
                               f_token=8, qtrans=True, num_feature_levels=4)
    model, _, _ = build_model(margs)
    model = model.to(dev).eval()
    ops.set_gemm_mode(args.gemm_mode)

    T, H, W = args.frames, args.height, args.width
    g = torch.Generator().manual_seed(1234 + rank)
    n_pool = 4
    clips = [torch.randn(T, 3, H, W, generator=g).to(dev) for _ in range(n_pool)]
    ids = torch.randint(3, 50264, (n_pool, 1, args.tokens), generator=g)
    ids[:, :, 0], ids[:, :, -1] = 0, 2
    ids = ids.to(dev)
    targets = [{"size": torch.tensor([H, W])}]
    gather_buf, pending = None, None

    C = max(1, args.clips_in_flight)
    streams = [torch.cuda.Stream(device=dev) for _ in range(C)] if C > 1 else None

    def step(i, gather=True):
        nonlocal gather_buf, pending
        if C == 1:
            out = model([clips[i % n_pool]], ids[i % n_pool], targets)
            local = out["pred_masks"]
        else:  # C independent B=1 forwards in flight, one per stream / replay slot
            cur = torch.cuda.current_stream()
            outs = []
            for c in range(C):
                streams[c].wait_stream(cur)
                with torch.cuda.stream(streams[c]):
                    outs.append(model([clips[(i * C + c) % n_pool]], ids[(i * C + c) % n_pool], targets, slot=c))
            for c in range(C):
                cur.wait_stream(streams[c])
            out = outs[0]
            local = torch.cat([o["pred_masks"] for o in outs], 0)
        if world > 1 and gather:
            # the masks of all world*C clips of this step meet on every rank; the collective of step i runs on RCCL's
            # stream while step i+1 computes, and is completed (stream-ordered) before step i+1's own gather starts
            if pending is not None:
                gather_buf = pending.wait()
            pending = gather_clip_masks_async(local if args.backend == "nccl" else local.cpu(), world * C)
        return out

    def fence():
        nonlocal gather_buf, pending
        if pending is not None:  # the last step's masks must have met inside the timed region
            gather_buf = pending.wait()
            pending = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    roofline = None
    C_saved, C = C, 1  # the instrumented pass and the parity check run one clip at a time
    if rank == 0 and not args.no_roofline:
        graph_mode, model.use_graph = model.use_graph, False  # per-launch events need eager launches
        step(0, gather=False)  # rank-0-only passes must not enter the collective
        ops.GEMM_PROFILE = []
        for i in range(args.steps):
            step(i, gather=False)
        torch.cuda.synchronize()
        prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
        model.use_graph = graph_mode
        agg = {}
        for tile, conv, flops, e0, e1 in prof:
            a = agg.setdefault((tile, conv), [0.0, 0.0, 0])
            a[0] += flops
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        key = max(agg, key=lambda k: agg[k][0])  # the kernel instance that carries the most algorithmic FLOPs
        fl, sec, n = agg[key]
        mode = ops.get_gemm_mode()
        # f32: v_mfma_f32_32x32x2_f32 (157.3 TFLOP/s).  f16x3: three v_mfma_f32_32x32x16_f16 per algorithmic product
        # (dense fp16 peak 2500 TFLOP/s): `achieved` stays ALGORITHMIC 2*M*N*K, the MFMA pipe issues 3x that.
        peak = 157.3 if mode == "f32" else 2500.0
        ach = fl / sec / 1e12
        gemm_sec_per_step = sum(v[1] for v in agg.values()) / args.steps
        tname = {256128: "256,128", 128128: "128,128", 12864: "128,64", 6464: "64,64", 6465: "64,64"}[key[0]]
        kname = "gemm_f32_kernel" if mode == "f32" else "gemm_f16x3_kernel"
        roofline = {"bound": "mfma", "kernel": f"{kname}<{tname},{'true' if key[1] else 'false'}>", "gemm_mode": mode,
                    "mfma_issued_tflops": round(ach * (1 if mode == "f32" else 3), 2),
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    # what a bare loop of this MFMA + its LDS fragment feed sustains with all 256 CUs issuing
                    # (tools/mfma_peak.py: clock-limited); informative, `peak` stays the guide's figure
                    "sustained_mfma_ceiling_tflops": None if mode == "f32" else 1650.0,
                    "traffic": _pmc_traffic(f"{kname}<{tname.replace(',', ', ')}"), "launches_per_step": n // args.steps,
                    "avg_launch_us": round(sec / n * 1e6, 2), "flops_per_launch_avg": fl / n,
                    "all_gemm_ms_per_step": round(gemm_sec_per_step * 1e3, 3),
                    "all_gemm_tflops": round(sum(v[0] for v in agg.values()) / sum(v[1] for v in agg.values()) / 1e12, 2)}

    cpu_baseline, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import tce_oracle as O
        cores = min(32, os.cpu_count() or 1)  # oneDNN/MKL scale poorly past ~32 threads on this graph
        torch.set_num_threads(cores)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if not k.startswith("text_encoder.")}
        frames_cpu = clips[0].cpu()
        with torch.no_grad():
            hid, pooled = model.forward_text_encoder(ids[0], dev)
            hid, pooled = hid.cpu(), pooled.cpu()
            cfg = O.OracleConfig(backbone=args.backbone)
            times = []
            for _ in range(2):
                t1 = time.perf_counter()
                ref = O.forward(sd, cfg, frames_cpu, hid, pooled, img_size=(H, W))
                times.append(time.perf_counter() - t1)
        cpu_baseline = {"value": round(1.0 / times[-1], 4), "unit": "clips/s", "cores": cores, "kind": "port",
                        "sample": f"1 clip (T={T}, {H}x{W}, {args.tokens} tokens), 2nd of 2 oracle forwards, "
                                  f"{times[-1]:.2f} s; text encoder excluded"}
        out = step(0, gather=False)
        torch.cuda.synchronize()
        pm = out["pred_masks"].cpu()
        parity = {"mask_iou_vs_oracle": round(O.mask_iou(pm > 0, ref["pred_masks"] > 0), 6),
                  "max_abs_logit_err": float((pm - ref["pred_masks"]).abs().max()),
                  # pixels where a sign flip is numerically meaningless
                  "frac_pixels_abs_logit_lt_1e-3": float((ref["pred_masks"].abs() < 1e-3).float().mean())}

    C = C_saved
    if rank == 0:
        known = {("resnet50", 1, 360, 640): "BASELINE config 1", ("swin_t_p4w7", 5, 360, 640): "BASELINE config 2",
                 ("video_swin_t_p4w7", 8, 384, 640): "BASELINE config 3", ("swin_b_p4w7", 10, 480, 854): "BASELINE config 5"}
        cfg_name = known.get((args.backbone, T, H, W), "not a BASELINE config")
        clips_total = args.steps * world * C
        line = {"metric": METRIC, "value": round(clips_total / elapsed, 3), "unit": "clips/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.gemm_mode == "f32" else "f32 (3xf16-split MFMA, f32 accumulate)", "data": "synthetic",
                "config": {"workload": f"{args.backbone} T={T} {H}x{W} + {args.tokens}-token text, B=1 clip per forward, "
                                       f"flags --with_box_refine --binary --f_token 8 --qtrans ({cfg_name})",
                           "clips_per_step": world * C, "clips_in_flight_per_gpu": C,
                           "parallelism": f"clip-sharded x{world}" +
                                                                   (" + RCCL all_gather(pred_masks)" if world > 1 else "")},
                "launch": "hipGraph replay" if model.use_graph else "eager",
                "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
