"""Thin tensor plumbing over the C ABI (include/tce_rvos.h): torch owns device memory and streams, every
arithmetic stage is a HIP kernel of libtce_rvos.so.  No op here has a PyTorch fallback."""
import ctypes as C

import os
import threading

import torch

from ._lib import GemmArgs, check, lib

ACT_NONE, ACT_RELU, ACT_GELU, ACT_RELU_AFTER_RES = 0, 1, 2, 3
RES_NONE, RES_ADD, RES_MUL = 0, 1, 2


class Arena:
    """Bump allocator over one device buffer with mark/release scoping.  A forward pass makes the same
    sequence of requests every time, so addresses are stable across calls -- what hipGraph replay needs."""

    ALIGN = 256
    FLAGS = 1 << 14

    def __init__(self, device, nbytes):
        self.device = torch.device(device)
        self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        self.off = 0
        self.peak = 0
        # zeroed now, outside any graph capture (an allocation inside a capture would come from the capture's private pool and its
        # fill would become a graph node)
        self.flags = torch.zeros(self.FLAGS, dtype=torch.int32, device=self.device)
        self.flag_off = 0

    def reset(self):
        self.off = 0
        self.flag_off = 0

    def alloc_flags(self, n):
        """n int32 counters that are ZERO now and that every kernel using them leaves zero (tce_ffn_fused_split_f32): a region of
        its own beside the bump buffer, handed out in request order like the buffer's bytes, never recycled inside a pass."""
        if self.flag_off + n > self.FLAGS:
            raise MemoryError("tce_rvos_amd arena: out of split counters")
        v = self.flags[self.flag_off:self.flag_off + n]
        self.flag_off += n
        return v

    def mark(self):
        return self.off

    def release(self, mark):
        self.off = mark

    def alloc(self, *shape, dtype=torch.float32):
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        start = (self.off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        end = start + nbytes
        if end > self.buf.numel():
            raise MemoryError(f"tce_rvos_amd arena exhausted: need {end} bytes, have {self.buf.numel()}")
        self.off = end
        self.peak = max(self.peak, end)
        return self.buf[start:end].view(dtype).view(*[int(s) for s in shape])


# When set to a list, every GEMM launch is bracketed by events on the launch stream and
# (tile, flops, start_event, end_event) is appended: bench.py's live roofline measurement.
GEMM_PROFILE = None
# Same for the HBM-bound kernels that report it: (kernel, rows, algorithmic bytes, start_event, end_event).
HBM_PROFILE = None


def _gemm_launch(g, splitk=1, ws=None, ln=None, eps=1e-5):
    def go():
        if splitk > 1:
            if ws is None or ws.numel() < splitk * g.M * g.N:
                raise ValueError("split-K GEMM needs a workspace of splits*M*N floats")
            if ln is not None:  # LayerNorm folded into the reduction pass
                check(lib().tce_gemm_splitk_ln_f32(C.byref(g), splitk, ws.data_ptr(), ln[0].data_ptr(), ln[1].data_ptr(), eps,
                                                   _stream()), "tce_gemm_splitk_ln_f32")
                return
            check(lib().tce_gemm_splitk_f32(C.byref(g), splitk, ws.data_ptr(), _stream()), "tce_gemm_splitk_f32")
        else:
            check(lib().tce_gemm_f32(C.byref(g), _stream()), "tce_gemm_f32")
    if GEMM_PROFILE is None:
        go()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    b = max(1, g.batch)
    GEMM_PROFILE.append((lib().tce_gemm_select_tile_ex(g.M, g.N, g.K, b, g.conv), bool(g.conv), 2.0 * g.M * g.N * g.K * b, e0, e1))


def set_gemm_mode(mode):
    """'f32' = exact fp32 MFMA; 'f16x3' (default) = fp32 operands split into two fp16 halves, 3 fp16 MFMAs per product;
    'f16' = one fp16 MFMA per product on operands rounded to fp16, fp32 accumulate (BASELINE config 5).  Packed weight
    streams carry the mode's rounding: call model.repack() after switching."""
    check(lib().tce_set_gemm_mode({"f32": 0, "f16x3": 1, "f16": 2}[mode]), "tce_set_gemm_mode")


def get_gemm_mode():
    """The calling thread's effective mode: its `arith` override if one is active, else the process default."""
    return {0: "f32", 1: "f16x3", 2: "f16"}[lib().tce_get_gemm_mode()]


_ARITH_TL = threading.local()


class arith:
    """`with ops.arith("f16"):` -- the launches (and weight packs) inside run in that arithmetic; None = leave the mode
    alone.  The mode is read on the host when a launch is issued (it is an argument of the kernels), so a captured
    hipGraph keeps, per node, the mode that was current at capture: this is how one clip mixes single-pass fp16 sites with
    split-fp16 ones (model.arith_policy, BASELINE config 5).  The override is per THREAD (tce_set_gemm_mode_thread): another
    host thread issuing launches meanwhile keeps its own arithmetic; set_gemm_mode sets the process default."""

    def __init__(self, mode):
        self.mode, self.prev = mode, None

    def __enter__(self):
        if self.mode is not None:
            self.prev = getattr(_ARITH_TL, "mode", None)
            _ARITH_TL.mode = self.mode
            check(lib().tce_set_gemm_mode_thread({"f32": 0, "f16x3": 1, "f16": 2}[self.mode]), "tce_set_gemm_mode_thread")
        return self

    def __exit__(self, *exc):
        if self.mode is not None:
            _ARITH_TL.mode = self.prev
            check(lib().tce_set_gemm_mode_thread(-1 if self.prev is None else {"f32": 0, "f16x3": 1, "f16": 2}[self.prev]),
                  "tce_set_gemm_mode_thread")
        return False


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------------------------------------------------------
# Operand-range guard of the split-fp16 arithmetic (include/tce_rvos.h: tce_set_range_flag).  fp16 saturates at
# 65504: weights are checked on the host when they are packed (check_weight_range), activations where they are
# produced -- every split-mode GEMM / fused-FFN epilogue raises a sticky device flag when it stores |v| >= 60000,
# Inf or NaN.  Reading the flag needs a host sync, so the model reads it at ITS sync points (one clip late in
# steady state); check_range() reads it now.
# ---------------------------------------------------------------------------------------------------------------
FP16_RANGE_LIMIT = 60000.0
_RANGE = {}


class RangeError(RuntimeError):
    pass


def range_flag(device=None):
    """The device flag (int32[1]) registered with the library; created (and registered) on first use."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _RANGE.get(dev)
    if st is None:
        st = {"flag": torch.zeros(1, dtype=torch.int32, device=dev), "host": torch.zeros(1, dtype=torch.int32).pin_memory(),
              "event": None}
        _RANGE[dev] = st
        # the library keeps one flag per device (filed under the device that owns the pointer): registered once
        check(lib().tce_set_range_flag(st["flag"].data_ptr()), "tce_set_range_flag")
    return st


def range_snapshot_async(device=None):
    """Queues a 4-byte copy of the flag to pinned host memory on the current stream (no host sync)."""
    st = range_flag(device)
    st["host"].copy_(st["flag"], non_blocking=True)
    st["event"] = torch.cuda.Event()
    st["event"].record()


def range_poll(device=None, wait=False):
    """Raises RangeError if a completed snapshot shows the flag set.  wait=True blocks on the pending snapshot."""
    st = range_flag(device)
    ev = st["event"]
    if ev is None:
        return
    if wait:
        ev.synchronize()
    elif not ev.query():
        return
    st["event"] = None
    if int(st["host"][0]) != 0:
        st["flag"].zero_()
        st["host"].zero_()
        raise RangeError("an activation left the fp16 range of the split-fp16 GEMM arithmetic (|x| >= 60000, Inf or NaN) in a "
                         "previous launch: its results are invalid.  Run with ops.set_gemm_mode('f32') (exact fp32 MFMA).")


def check_range(device=None):
    """Synchronous form: snapshot the flag now, wait, raise RangeError if any launch so far tripped it."""
    range_snapshot_async(device)
    range_poll(device, wait=True)


def check_weight_range(named_tensors):
    """Host-side half of the guard, run when operands are packed: every GEMM weight must lie inside the fp16 range."""
    bad = []
    for name, t in named_tensors:
        if t.is_floating_point() and t.numel() and not bool(t.abs().max() < FP16_RANGE_LIMIT):
            bad.append(name)
    if bad and get_gemm_mode() != "f32":
        raise RangeError(f"parameters outside the fp16 range of the split-fp16 GEMM arithmetic (|w| >= {FP16_RANGE_LIMIT:g} or "
                         f"non-finite): {bad[:5]}{'...' if len(bad) > 5 else ''}.  Select the exact fp32 kernels with "
                         f"ops.set_gemm_mode('f32') before loading these weights.")


def _hbm_timed(name, rows, nbytes, go):
    """Runs go(); with HBM_PROFILE on, brackets it with events on the launch stream and records its algorithmic bytes."""
    if HBM_PROFILE is None:
        go()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    HBM_PROFILE.append((name, rows, float(nbytes), e0, e1))


def _chk(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f"{name}: expected a CUDA float32 tensor, got {t.dtype} on {t.device}")


def _rows(t, name):
    """2-D row view: returns (ptr, rows, cols, ld).  Accepts [..., C] contiguous or 2-D with row stride."""
    _chk(t, name)
    if t.dim() == 2 and t.stride(1) == 1:
        return t.data_ptr(), t.shape[0], t.shape[1], t.stride(0)
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous or a 2-D row-strided view")
    c = t.shape[-1]
    return t.data_ptr(), t.numel() // c, c, c


def gemm(a, w, bias=None, a2=None, act=ACT_NONE, res=None, res_mode=RES_NONE, out=None, alloc=None):
    """out[M,N] = epi((a (+a2)) @ w.T + bias).  a [M,K] (row-strided ok), w [N,K]."""
    ap, M, K, lda = _rows(a, "a")
    wp, N, Kw, ldw = _rows(w, "w")
    if K != Kw:
        raise ValueError(f"gemm: K mismatch {K} vs {Kw}")
    if out is None:
        out = alloc(M, N) if alloc else torch.empty(M, N, dtype=torch.float32, device=a.device)
    op, Mo, No, ldc = _rows(out, "out")
    if Mo != M or No != N:
        raise ValueError("gemm: out shape mismatch")
    g = GemmArgs()
    g.A, g.W, g.C = ap, wp, op
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldw, g.ldc = lda, ldw, ldc
    if a2 is not None:
        p2, M2, K2, ld2 = _rows(a2, "a2")
        if M2 != M or K2 != K:
            raise ValueError("gemm: a2 shape mismatch")
        g.A2, g.lda2 = p2, ld2
    if bias is not None:
        _chk(bias, "bias")
        g.bias = bias.data_ptr()
    if res_mode != RES_NONE:
        rp, Mr, Nr, ldr = _rows(res, "res")
        if Mr != M or Nr != N:
            raise ValueError("gemm: res shape mismatch")
        g.res, g.ldres = rp, ldr
    g.act, g.res_mode, g.batch = act, res_mode, 1
    _gemm_launch(g)
    return out


def gemm_batched(a, w, out, bias=None, act=ACT_NONE):
    """a [B,M,K], w [B,N,K], out [B,M,N] (all contiguous): B independent problems in one launch."""
    for t, n in ((a, "a"), (w, "w"), (out, "out")):
        _chk(t, n)
        if not t.is_contiguous() or t.dim() != 3:
            raise ValueError(f"gemm_batched: {n} must be contiguous 3-D")
    B, M, K = a.shape
    _, N, _ = w.shape
    g = GemmArgs()
    g.A, g.W, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldw, g.ldc = K, K, N
    g.batch = B
    g.sA, g.sW, g.sC = M * K, N * K, M * N
    if bias is not None:
        g.bias, g.sBias = bias.data_ptr(), N
    g.act = act
    _gemm_launch(g)
    return out


def conv2d_cl(x, w_packed, T, H, W, Cin, kh, kw, stride, pad, bias=None, act=ACT_NONE, out=None, alloc=None, res=None,
              res_mode=RES_NONE, splitk=1, ws=None):
    """Channels-last convolution as implicit GEMM.  x [T*H*W, Cin]; w_packed [N, kh*kw*Cin] with
    k = (ky*kw+kx)*Cin + c.  Returns [T*Ho*Wo, N]."""
    _chk(x, "x")
    _chk(w_packed, "w")
    N = w_packed.shape[0]
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    M = T * Ho * Wo
    if (kh == 3 and kw == 3 and stride == 1 and pad == 1 and act == ACT_NONE and res_mode == RES_NONE and splitk == 1
            and M >= CONV3_MIN_PIXELS and get_gemm_mode() != "f32"):
        pk = _ROUTES.conv3.get((w_packed.data_ptr(), N, w_packed.shape[1], get_gemm_mode()))
        if pk is not None and x.stride(0) % 4 == 0:
            return conv3x3(x, pk, T, H, W, Cin, N, bias=bias, out=out, alloc=alloc), Ho, Wo
    if out is None:
        out = alloc(M, N) if alloc else torch.empty(M, N, dtype=torch.float32, device=x.device)
    g = GemmArgs()
    g.A, g.W, g.C = x.data_ptr(), w_packed.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, kh * kw * Cin
    g.lda, g.ldw, g.ldc = Cin, kh * kw * Cin, N
    if bias is not None:
        g.bias = bias.data_ptr()
    if res_mode != RES_NONE:
        _chk(res, "res")
        g.res, g.ldres = res.data_ptr(), N
    g.act, g.res_mode, g.batch, g.conv = act, res_mode, 1, 1
    g.T, g.H, g.Wd, g.Cin, g.Ho, g.Wo = T, H, W, Cin, Ho, Wo
    g.kh, g.kw, g.stride, g.pad = kh, kw, stride, pad
    _gemm_launch(g, splitk, ws)
    return out, Ho, Wo


def layernorm(x, gamma, beta, eps=1e-5, r=None, out=None, alloc=None):
    _chk(x, "x")
    Cn = x.shape[-1]
    M = x.numel() // Cn
    if out is None:
        out = alloc(*x.shape) if alloc else torch.empty_like(x)
    check(lib().tce_layernorm_f32(x.data_ptr(), r.data_ptr() if r is not None else None, gamma.data_ptr(),
                                  beta.data_ptr(), out.data_ptr(), M, Cn, eps, _stream()), "tce_layernorm_f32")
    return out


def groupnorm_cl(x, gamma, beta, T, HW, Cn, G, eps=1e-5, relu=False, out=None, ws=None, alloc=None):
    """x [T*HW, C] channels-last."""
    _chk(x, "x")
    nsplit = lib().tce_groupnorm_nsplit(HW)
    if ws is None:
        ws = alloc(T * G * (nsplit * 3 + 2)) if alloc else torch.empty(T * G * (nsplit * 3 + 2), dtype=torch.float32, device=x.device)
    if out is None:
        out = alloc(T * HW, Cn) if alloc else torch.empty(T * HW, Cn, dtype=torch.float32, device=x.device)
    # algorithmic bytes: the map read once for the statistics, once for the apply pass, written once
    _hbm_timed("groupnorm (stats + apply)", T * HW, 3.0 * T * HW * Cn * 4, lambda: check(
        lib().tce_groupnorm_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), ws.data_ptr(), T, HW, Cn, G, eps,
                                1 if relu else 0, _stream()), "tce_groupnorm_f32"))
    return out


GN_UP_FUSE = os.environ.get("TCE_GN_UP_FUSE", "1") != "0"  # A/B: 0 = GroupNorm apply and the top-down merge as two launches


def groupnorm_up_add(x, gamma, beta, T, h, w, ho, wo, Cn, G, add, out, eps=1e-5, relu=True, ws=None, alloc=None):
    """out <- add + act(GN(x)) up-sampled (nearest) from [T, h, w, C] onto [T, ho, wo, C] (tce_groupnorm_up_add_f32)."""
    _chk(x, "x")
    nsplit = lib().tce_groupnorm_nsplit(h * w)
    if ws is None:
        ws = alloc(T * G * (nsplit * 3 + 2)) if alloc else torch.empty(T * G * (nsplit * 3 + 2), dtype=torch.float32, device=x.device)
    check(lib().tce_groupnorm_up_add_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), add.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                         T, h, w, ho, wo, Cn, G, eps, 1 if relu else 0, _stream()), "tce_groupnorm_up_add_f32")
    return out


def resnet_stem(frames, w_k64, bias, out=None, alloc=None):
    """conv 7x7/s2/p3 (3 -> 64) + folded FrozenBatchNorm2d + ReLU; frames NCHW -> channels-last [T*Ho*Wo, 64]."""
    _chk(frames, "frames")
    T, c3, H, W = frames.shape
    if c3 != 3 or not frames.is_contiguous():
        raise ValueError("resnet_stem: frames must be contiguous [T,3,H,W]")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = alloc(T * Ho * Wo, 64) if alloc else torch.empty(T * Ho * Wo, 64, dtype=torch.float32, device=frames.device)
    check(lib().tce_resnet_stem_f32(frames.data_ptr(), w_k64.data_ptr(), bias.data_ptr(), out.data_ptr(), T, H, W,
                                    _stream()), "tce_resnet_stem_f32")
    return out, Ho, Wo


def maxpool3x3s2_cl(x, T, H, W, Cn, out=None, alloc=None):
    _chk(x, "x")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = alloc(T * Ho * Wo, Cn) if alloc else torch.empty(T * Ho * Wo, Cn, dtype=torch.float32, device=x.device)
    check(lib().tce_maxpool3x3s2_cl_f32(x.data_ptr(), out.data_ptr(), T, H, W, Cn, _stream()), "tce_maxpool3x3s2_cl_f32")
    return out, Ho, Wo


def patch_embed(frames, w, b, gamma, beta, eps=1e-5, out=None, alloc=None):
    _chk(frames, "frames")
    T, c3, H, W = frames.shape
    if c3 != 3 or not frames.is_contiguous():
        raise ValueError("patch_embed: frames must be contiguous [T,3,H,W]")
    Cn = w.shape[0]
    Hp, Wp = (H + 3) // 4, (W + 3) // 4
    if out is None:
        out = alloc(T * Hp * Wp, Cn) if alloc else torch.empty(T * Hp * Wp, Cn, dtype=torch.float32, device=frames.device)
    _hbm_timed("patch_embed", T * Hp * Wp, 4.0 * (T * 3 * H * W + T * Hp * Wp * Cn), lambda: check(
        lib().tce_patch_embed_f32(frames.data_ptr(), w.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                  T, H, W, Cn, eps, _stream()), "tce_patch_embed_f32"))
    return out, Hp, Wp


def window_attn(qkv, qkv_bias, table, T, H, W, Cn, nH, shift, out=None, alloc=None):
    _chk(qkv, "qkv")
    if out is None:
        out = alloc(T * H * W, Cn) if alloc else torch.empty(T * H * W, Cn, dtype=torch.float32, device=qkv.device)
    # algorithmic bytes: qkv in (3C per token) + out (C per token)
    _hbm_timed("window_attn", T * H * W, 4.0 * T * H * W * 4 * Cn, lambda: check(
        lib().tce_window_attn_f32(qkv.data_ptr(), qkv_bias.data_ptr(), table.data_ptr(), out.data_ptr(), T, H, W, Cn, nH, shift,
                                  _stream()), "tce_window_attn_f32"))
    return out


def window_attn3d(qkv, qkv_bias, table, T, H, W, Cn, nH, shifted, out=None, alloc=None):
    _chk(qkv, "qkv")
    if out is None:
        out = alloc(T * H * W, Cn) if alloc else torch.empty(T * H * W, Cn, dtype=torch.float32, device=qkv.device)
    check(lib().tce_window_attn3d_f32(qkv.data_ptr(), qkv_bias.data_ptr(), table.data_ptr(), out.data_ptr(), T, H, W,
                                      Cn, nH, 1 if shifted else 0, _stream()), "tce_window_attn3d_f32")
    return out


def patch_merge_ln(x, gamma, beta, T, H, W, Cn, eps=1e-5, out=None, alloc=None):
    _chk(x, "x")
    H2, W2 = (H + 1) // 2, (W + 1) // 2
    if out is None:
        out = alloc(T * H2 * W2, 4 * Cn) if alloc else torch.empty(T * H2 * W2, 4 * Cn, dtype=torch.float32, device=x.device)
    check(lib().tce_patch_merge_ln_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), T, H, W, Cn, eps,
                                       _stream()), "tce_patch_merge_ln_f32")
    return out, H2, W2


# ---------------------------------------------------------------------------------------------------------------
# Swin attention half-block as one launch (csrc/swinattn.hip): x <- x + proj(window_attention(norm1(x)))
# ---------------------------------------------------------------------------------------------------------------
SWIN_FUSED_C = (96, 128, 192, 256)
SWIN_FUSED = os.environ.get("TCE_SWIN_FUSED", "1") != "0"  # A/B: 0 = norm1->qkv, window attention, proj+res as three launches


def swin_attn_pack(wqkv, wproj):
    """Wqkv [3C, C], Wproj [C, C] (nn.Linear layouts) -> the kernel's weight stream, in the CURRENT arithmetic."""
    _chk(wqkv, "wqkv")
    _chk(wproj, "wproj")
    Cn = wproj.shape[0]
    nbytes = lib().tce_swin_attn_packed_bytes(Cn)
    if nbytes < 0 or tuple(wqkv.shape) != (3 * Cn, Cn) or tuple(wproj.shape) != (Cn, Cn):
        raise ValueError(f"swin_attn_pack: unsupported shapes {tuple(wqkv.shape)}, {tuple(wproj.shape)}")
    out = torch.empty(nbytes, dtype=torch.uint8, device=wqkv.device)
    check(lib().tce_swin_attn_pack_f32(wqkv.contiguous().data_ptr(), wproj.contiguous().data_ptr(), out.data_ptr(), Cn, _stream()),
          "tce_swin_attn_pack_f32")
    return out


def swin_attn_fused(x, pk, qkv_bias, proj_bias, table, g1, b1, T, H, W, Cn, shift, eps=1e-5, out=None):
    """out (default: x, in place) = x + proj(window_attention(LayerNorm(x)));  x [T*H*W, C] token-major."""
    _chk(x, "x")
    if out is None:
        out = x
    ldx = x.stride(0) if x.dim() == 2 else Cn
    ldo = out.stride(0) if out.dim() == 2 else Cn

    def go():
        check(lib().tce_swin_attn_fused_f32(x.data_ptr(), ldx, pk.data_ptr(), qkv_bias.data_ptr(), proj_bias.data_ptr(), table.data_ptr(),
                                            g1.data_ptr(), b1.data_ptr(), eps, out.data_ptr(), ldo, T, H, W, Cn, shift, _stream()),
              "tce_swin_attn_fused_f32")
    if GEMM_PROFILE is None:
        go()
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    nwin = T * ((H + 6) // 7) * ((W + 6) // 7)
    # algorithmic FLOPs: qkv + proj on the real tokens, scores + AV on 49 x 49 per (window, head)
    GEMM_PROFILE.append((f"swin_attn_fused_kernel<{Cn}", False, 8.0 * T * H * W * Cn * Cn + 4.0 * nwin * 49 * 49 * Cn, e0, e1))
    return out


_MHA_SPLIT_OFF = os.environ.get("TCE_MHA_SPLIT", "1") == "0"  # A/B: every attention launch on the exact fp32-MFMA kernel
_mha_split_applied = False


def _apply_mha_split():
    """The A/B switch is applied at the first attention launch, not at import: importing the package must not load (or
    require) the HIP library -- bench.py builds it on a fresh checkout AFTER importing the package."""
    global _mha_split_applied
    if not _mha_split_applied:
        _mha_split_applied = True
        if _MHA_SPLIT_OFF:
            lib().tce_debug_mha_set_split(0)


MHA_WS_MIN_KEYS = int(os.environ.get("TCE_MHA_WS_MIN_KEYS", 1024))


def mha_core(q, k, v, batch, nheads, Lq, Lk, ldq, ldk, ldv, sQ, sK, sV, out, ldo, sO, kmask=None, scale=None, alloc=None):
    """Raw-strided attention core; q/k/v/out are tensors whose data_ptr() is the first element of
    (batch 0, row 0, head 0).  Strides in floats.  alloc: long key sequences take the pre-split form (tce_mha_ws_f32),
    whose fp16 planes of K / V come from it."""
    if scale is None:
        scale = 32 ** -0.5
    _apply_mha_split()
    if alloc is not None and Lk >= MHA_WS_MIN_KEYS and get_gemm_mode() != "f32":
        ws = alloc(lib().tce_mha_ws_bytes(batch, nheads, Lk) // 4)
        check(lib().tce_mha_ws_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), ws.data_ptr(), batch, nheads, Lq, Lk,
                                   ldq, ldk, ldv, ldo, sQ, sK, sV, sO, kmask.data_ptr() if kmask is not None else None, scale,
                                   _stream()), "tce_mha_ws_f32")
        return out
    check(lib().tce_mha_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), batch, nheads, Lq, Lk, ldq, ldk,
                            ldv, ldo, sQ, sK, sV, sO, kmask.data_ptr() if kmask is not None else None, scale,
                            _stream()), "tce_mha_f32")
    return out


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    """Drop-in for MultiScaleDeformableAttention_update.ms_deform_attn_forward (ms_deform_attn_func.py:26-27)."""
    for t, n in ((value, "value"), (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight")):
        _chk(t, n)
        if not t.is_contiguous():
            raise RuntimeError(f"{n} tensor has to be contiguous")
    if not (spatial_shapes.is_cuda and level_start_index.is_cuda and spatial_shapes.dtype == torch.int64):
        raise RuntimeError("spatial_shapes / level_start_index must be CUDA int64 tensors")
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    out = torch.empty(N, Lq, M * D, dtype=torch.float32, device=value.device)
    check(lib().tce_ms_deform_attn_forward_f32(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                               sampling_loc.data_ptr(), attn_weight.data_ptr(), out.data_ptr(), N, S,
                                               M, D, Lq, L, P, _stream()), "tce_ms_deform_attn_forward_f32")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step=64):
    """Drop-in for MultiScaleDeformableAttention_update.ms_deform_attn_backward (ms_deform_attn_func.py:36-41):
    returns (grad_value, grad_sampling_loc, grad_attn_weight)."""
    for t, n in ((value, "value"), (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight"), (grad_output, "grad_output")):
        _chk(t, n)
        if not t.is_contiguous():
            raise RuntimeError(f"{n} tensor has to be contiguous")
    if not (spatial_shapes.is_cuda and level_start_index.is_cuda and spatial_shapes.dtype == torch.int64):
        raise RuntimeError("spatial_shapes / level_start_index must be CUDA int64 tensors")
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    if grad_output.numel() != N * Lq * M * D:
        raise RuntimeError("grad_output must be [N, Lq, M*D]")
    gv, gl, ga = torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
    check(lib().tce_ms_deform_attn_backward_f32(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                                sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                                                gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), N, S, M, D, Lq, L, P, _stream()),
          "tce_ms_deform_attn_backward_f32")
    return gv, gl, ga


class MSDeformAttnFunction(torch.autograd.Function):
    """The reference's autograd wrapper (models/ops/functions/ms_deform_attn_func.py:20-43) over the HIP op: same
    argument list, same returned gradients (value, -, -, sampling_locations, attention_weights, -)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step=64, is_3d=False):
        if is_3d:
            raise NotImplementedError("MSDeformAttnFunction: is_3d=True (the reference's 3-D sampling variant) is not built; "
                                      "the per-clip path only uses the 2-D op")
        ctx.im2col_step = im2col_step
        out = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                     attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, aw = ctx.saved_tensors
        gv, gl, ga = ms_deform_attn_backward(value, shapes, lsi, loc, aw, grad_output.contiguous(), ctx.im2col_step)
        return gv, None, None, gl, ga, None, None


def msda_fused(value, proj, ref, shapes_hw, N, S, M, Lq, L, P, ref_dim, ref_per_frame, out=None, alloc=None, valid_hw=None):
    """valid_hw: per level (rows, columns) that are not padding (padded clips; None = un-padded)."""
    _chk(value, "value")
    if out is None:
        out = alloc(N * Lq, M * 32) if alloc else torch.empty(N * Lq, M * 32, dtype=torch.float32, device=value.device)
    arr = (C.c_int32 * (2 * L))(*[int(v) for hw in shapes_hw for v in hw])
    varr = (C.c_int32 * (2 * L))(*[int(v) for hw in valid_hw for v in hw]) if valid_hw is not None else None

    def go():
        check(lib().tce_msda_fused_valid_f32(value.data_ptr(), proj.data_ptr(), ref.data_ptr(), out.data_ptr(), arr, varr, N, S,
                                             M, Lq, L, P, ref_dim, 1 if ref_per_frame else 0, _stream()), "tce_msda_fused_f32")
    if HBM_PROFILE is None:
        go()
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    # algorithmic bytes: value read once + the 384-wide offsets|weights projection + the output rows
    HBM_PROFILE.append(("msda_fused_q4u_kernel", N * Lq, 4.0 * (N * S * M * 32 + N * Lq * M * L * P * 3 + N * Lq * M * 32), e0, e1))
    return out


def msda_fewq_raw(src, wv, bv, proj, ref, shapes_hw, N, S, Lq, L, P, ref_dim, ref_per_frame, out=None, alloc=None, valid_hw=None):
    """The attention core on the UN-projected rows src [N*S, 256] with the module's value_proj (wv [256,256], bv [256]) applied to
    the bilinear sample (csrc/msda.hip, tce_msda_fewq_raw_f32): what msda_fused returns on value_proj(src), for calls with a few
    queries per frame (no [N*S, 256] x [256, 256] projection)."""
    _chk(src, "src")
    if out is None:
        out = alloc(N * Lq, 256) if alloc else torch.empty(N * Lq, 256, dtype=torch.float32, device=src.device)
    arr = (C.c_int32 * (2 * L))(*[int(v) for hw in shapes_hw for v in hw])
    varr = (C.c_int32 * (2 * L))(*[int(v) for hw in valid_hw for v in hw]) if valid_hw is not None else None
    check(lib().tce_msda_fewq_raw_f32(src.data_ptr(), wv.data_ptr(), bv.data_ptr(), proj.data_ptr(), ref.data_ptr(), out.data_ptr(),
                                      arr, varr, N, S, 8, Lq, L, P, ref_dim, 1 if ref_per_frame else 0, _stream()),
          "tce_msda_fewq_raw_f32")
    return out


MSDA_RAW = os.environ.get("TCE_MSDA_RAW", "1") != "0"  # A/B: 0 = value_proj GEMM of the frame + the projected few-query form


def pos_sine2d(T, h, w, F, device, add=None, out=None, alloc=None, valid=None):
    """valid = (hv, wv): rows / columns of the map that are not padding (padded clips)."""
    if out is None:
        out = alloc(T * h * w, 2 * F) if alloc else torch.empty(T * h * w, 2 * F, dtype=torch.float32, device=device)
    hv, wv = (h, w) if valid is None else valid
    check(lib().tce_pos_sine2d_valid_f32(out.data_ptr(), add.data_ptr() if add is not None else None, T, h, w, F, hv, wv,
                                         _stream()), "tce_pos_sine2d_f32")
    return out


def resize_nearest(x, T, h, w, ho, wo, Cn, add=None, out=None, alloc=None):
    _chk(x, "x")
    if out is None:
        out = alloc(T * ho * wo, Cn) if alloc else torch.empty(T * ho * wo, Cn, dtype=torch.float32, device=x.device)
    check(lib().tce_resize_nearest_f32(x.data_ptr(), add.data_ptr() if add is not None else None, out.data_ptr(), T, h,
                                       w, ho, wo, Cn, _stream()), "tce_resize_nearest_f32")
    return out


RESIZE_LN_FUSE = os.environ.get("TCE_RESIZE_LN_FUSE", "1") != "0"  # A/B: 0 = bilinear resize + add and the LayerNorm as two launches


def resize_bilinear(x, T, h, w, ho, wo, Cn, add=None, out=None, alloc=None, ln=None, eps=1e-5):
    """ln = (gamma, beta): out = LayerNorm(add + bilinear(x)) in one pass (tce_resize_bilinear_ln_f32; C = 256, add required)."""
    _chk(x, "x")
    if out is None:
        out = alloc(T * ho * wo, Cn) if alloc else torch.empty(T * ho * wo, Cn, dtype=torch.float32, device=x.device)
    if ln is not None:
        if Cn == 256 and add is not None and RESIZE_LN_FUSE:
            check(lib().tce_resize_bilinear_ln_f32(x.data_ptr(), add.data_ptr(), ln[0].data_ptr(), ln[1].data_ptr(), eps, out.data_ptr(),
                                                   T, h, w, ho, wo, Cn, _stream()), "tce_resize_bilinear_ln_f32")
            return out
        resize_bilinear(x, T, h, w, ho, wo, Cn, add=add, out=out)
        return layernorm(out, ln[0], ln[1], eps, out=out)
    check(lib().tce_resize_bilinear_f32(x.data_ptr(), add.data_ptr() if add is not None else None, out.data_ptr(), T, h,
                                        w, ho, wo, Cn, _stream()), "tce_resize_bilinear_f32")
    return out


def add(a, b, out=None, alloc=None):
    """out = a + b with b broadcast over leading dims (b.numel() divides a.numel())."""
    _chk(a, "a")
    if out is None:
        out = alloc(*a.shape) if alloc else torch.empty_like(a)
    check(lib().tce_add_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), b.numel(), _stream()), "tce_add_f32")
    return out


def _copy_seg(src):
    """(rows, row_words, src_pitch_words) of a tensor the segment copy can move, or None."""
    es = src.element_size()
    if es % 4 or src.numel() == 0:
        return None
    k = es // 4
    if src.is_contiguous():
        return 1, src.numel() * k, src.numel() * k
    if src.dim() < 2 or src.stride(-1) != 1:
        return None
    pitch = src.stride(-2)
    exp = pitch * src.shape[-2]
    for d in range(src.dim() - 3, -1, -1):  # leading dims must collapse onto the row pitch
        if src.shape[d] != 1 and src.stride(d) != exp:
            return None
        exp *= src.shape[d] if src.shape[d] != 1 else 1
    return src.numel() // src.shape[-1], src.shape[-1] * k, pitch * k


class CopyPlan:
    """A fixed list of source tensors copied out with one launch per 16 (tce_copy_segments): `clone()` returns fresh
    dense tensors carved from one allocation."""

    def __init__(self, srcs):
        from ._lib import CopySeg
        self.srcs = list(srcs)
        self.geo = [_copy_seg(t) for t in self.srcs]
        if any(g is None for g in self.geo):
            raise ValueError("copy plan: a source is neither dense nor a uniform row gather of 4-byte words")
        self.offs, off = [], 0
        for t in self.srcs:
            self.offs.append(off)
            off += (t.numel() * t.element_size() + 255) // 256 * 256
        self.total = off
        self.segs = (CopySeg * len(self.srcs))()
        for sg, t, (rows, rw, pitch) in zip(self.segs, self.srcs, self.geo):
            sg.src, sg.rows, sg.row_words, sg.src_pitch_words = t.data_ptr(), rows, rw, pitch

    def _launch(self):
        from ._lib import CopySeg
        n, st = len(self.srcs), _stream()
        for lo in range(0, n, 16):
            m = min(16, n - lo)
            check(lib().tce_copy_segments(C.cast(C.byref(self.segs, lo * C.sizeof(CopySeg)), C.POINTER(CopySeg)), m, st),
                  "tce_copy_segments")

    def clone(self):
        buf = torch.empty(self.total, dtype=torch.uint8, device=self.srcs[0].device)
        base, outs = buf.data_ptr(), []
        for sg, t, off in zip(self.segs, self.srcs, self.offs):
            sg.dst = base + off
            outs.append(buf[off:off + t.numel() * t.element_size()].view(t.dtype).view(t.shape))
        self._launch()
        return outs


def copy_many(dsts, srcs):
    """dsts[i] <- srcs[i] (dense destinations of the same shape and dtype) in one launch per 16 tensors; tensors the
    segment copy cannot express go through Tensor.copy_."""
    from ._lib import CopySeg
    segs, n = (CopySeg * 16)(), 0
    st = _stream()

    def flush():
        nonlocal n
        if n:
            check(lib().tce_copy_segments(segs, n, st), "tce_copy_segments")
        n = 0

    for d, t in zip(dsts, srcs):
        geo = _copy_seg(t) if (d.shape == t.shape and d.dtype == t.dtype and d.is_contiguous() and
                               d.device == t.device) else None
        if geo is None:
            d.copy_(t)
            continue
        sg = segs[n]
        sg.src, sg.dst, sg.rows, sg.row_words, sg.src_pitch_words = t.data_ptr(), d.data_ptr(), geo[0], geo[1], geo[2]
        n += 1
        if n == 16:
            flush()
    flush()


def tile(src, reps, out=None, alloc=None):
    """out = src repeated `reps` times along a new leading axis (flattened)."""
    _chk(src, "src")
    if out is None:
        out = alloc(reps * src.numel()) if alloc else torch.empty(reps * src.numel(), dtype=torch.float32, device=src.device)
    check(lib().tce_tile_f32(src.data_ptr(), out.data_ptr(), src.numel(), reps, _stream()), "tce_tile_f32")
    return out


def sigmoid(x, out=None, alloc=None):
    if out is None:
        out = alloc(*x.shape) if alloc else torch.empty_like(x)
    check(lib().tce_sigmoid_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "tce_sigmoid_f32")
    return out


def box_refine(tmp, ref, out=None, alloc=None):
    n = tmp.numel() // 4
    ref_dim = ref.shape[-1]
    if out is None:
        out = alloc(*tmp.shape) if alloc else torch.empty_like(tmp)
    check(lib().tce_box_refine_f32(tmp.data_ptr(), ref.data_ptr(), out.data_ptr(), n, ref_dim, _stream()),
          "tce_box_refine_f32")
    return out


def contrastive(memory, sent, T, S, Cn, frames_per_clip, out, ws):
    """out[t] = cos(mean_S memory[t], sent[t // frames_per_clip]) (contrastive_cal, tce_rvos.py:512-521)."""
    check(lib().tce_contrastive_f32(memory.data_ptr(), sent.data_ptr(), out.data_ptr(), ws.data_ptr(), T, S, Cn, frames_per_clip,
                                    _stream()), "tce_contrastive_f32")
    return out


def mask_pack(params, nl, T, Q, Cm, w0f, tail):
    check(lib().tce_mask_pack_f32(params.data_ptr(), w0f.data_ptr(), tail.data_ptr(), nl, T, Q, Cm, _stream()),
          "tce_mask_pack_f32")


def mask_tail(G, tail, refs, ref_ld, masks, nl, T, Q, h, w, img_h, img_w, stride_px=4):
    check(lib().tce_mask_tail_f32(G.data_ptr(), tail.data_ptr(), refs.data_ptr(), ref_ld, masks.data_ptr(), nl, T, Q, h,
                                  w, float(img_h), float(img_w), stride_px, _stream()), "tce_mask_tail_f32")
    return masks


def gemm_ex(a, w, out, M, N, K, lda, ldw, ldc, bias=None, a2=None, lda2=0, act=ACT_NONE, res=None, ldres=0,
            res_mode=RES_NONE, batch=1, sA=0, sA2=0, sW=0, sBias=0, sC=0, sRes=0, splitk=1, ws=None, ln=None, ln_eps=1e-5):
    """Fully explicit form: tensors only provide base pointers (slices / views welcome); all sizes and
    strides (in floats) are given by the caller.  Used by the model for frame-batched launches where the
    addend (a positional map) is shared by all frames (sA2 = 0) or the output is a level slice of [T,S,C]."""
    if _ROUTES.rowlin and splitk <= 1 and ln is None and sW == 0 and sBias == 0 and act in (ACT_NONE, ACT_RELU, ACT_GELU):
        pk = _rowlin_route(w, M, N, K, ldw, batch)
        if pk is not None:
            return rowlin(a, pk, out, M, N, K, lda, ldc, bias=bias, a2=a2, lda2=lda2, act=act, res=res, ldres=ldres,
                          res_mode=res_mode, batch=batch, sX=sA, sA2=sA2, sRes=sRes, sOut=sC)
    g = GemmArgs()
    g.A, g.W, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldw, g.ldc = lda, ldw, ldc
    if a2 is not None:
        g.A2, g.lda2 = a2.data_ptr(), lda2
    if bias is not None:
        g.bias = bias.data_ptr()
    if res_mode != RES_NONE:
        g.res, g.ldres = res.data_ptr(), ldres
    g.act, g.res_mode, g.batch = act, res_mode, batch
    g.sA, g.sA2, g.sW, g.sBias, g.sC, g.sRes = sA, sA2, sW, sBias, sC, sRes
    if ln is not None and splitk <= 1:  # ln = the LayerNorm that follows: folded into a split-K reduction, else its own launch
        if ldc != N or batch != 1:
            raise ValueError("gemm_ex(ln=...): dense [M, N] output of one problem")
        _gemm_launch(g, splitk, ws)
        check(lib().tce_layernorm_f32(out.data_ptr(), None, ln[0].data_ptr(), ln[1].data_ptr(), out.data_ptr(), M, N, ln_eps,
                                      _stream()), "tce_layernorm_f32")
        return out
    _gemm_launch(g, splitk, ws, ln, ln_eps)
    return out


def _rowlin_route(w, M, N, K, ldw, batch):
    """Packed copy of a registered weight when the token-stationary kernel measured faster than the tiled GEMM for this
    shape (tools/rowlin_bench.py): K = 96 / 128 with many rows (Swin stage 1, the stride-4 adapter), and K = 192 / 256
    with N >= 384 (the 384-wide offset/weight projection, the 512-wide q|k projection)."""
    if K not in ROWLIN_K or N % 32 or get_gemm_mode() == "f32":
        return None
    rows = M * max(1, batch)
    if K == 384:  # Swin-T stage 3 (4600 rows at config 2, 7680 at config 3): the tiled GEMM's 12 K slices per tile are latency-
        # bound there; from ~12000 rows on the tiled kernel wins again (18000 rows: 94 vs 104 us, tools/rowlin384_bench.py)
        if rows < 2048 or M < 2048 or N < 384 or rows > ROWLIN384_MAX_ROWS:
            return None
        return _ROUTES.rowlin.get((w.data_ptr(), N, K, ldw, get_gemm_mode()))
    if K == 512:  # Swin-B stage 3 (16200 rows at config 5): one wave per SIMD, so a tile's epilogue is not hidden -- only the
        # proj + residual launch wins (48 vs 56 us; norm1 -> qkv 117 vs 121, norm2 -> fc1 + GELU 184 vs 172 us: left on the tiled
        # kernel; profiles/r04_rowlin_k512.txt)
        if N != 512 or rows < ROWLIN_MIN_ROWS or M < 2048:
            return None
        return _ROUTES.rowlin.get((w.data_ptr(), N, K, ldw, get_gemm_mode()))
    if rows < ROWLIN_MIN_ROWS or M < 2048:
        return None
    if not (K <= 128 and rows >= 32768) and not (K >= 192 and N >= 384 and N != 576):
        return None
    return _ROUTES.rowlin.get((w.data_ptr(), N, K, ldw, get_gemm_mode()))


def rowlin_lookup(w, N, K, ldw=None):
    mode = get_gemm_mode()
    return _ROUTES.rowlin.get((w.data_ptr(), N, K, K if ldw is None else ldw, mode)) if mode != "f32" else None


# Split-K for the skinny, deep GEMMs of the side branches (RoBERTa at 32 tokens, the decoder on 25 rows); TCE_SPLITK=0
# disables.  Round 1 measured no gain (the branches were hidden behind a slower main chain); with the round-2 main
# chain (fused FFN / cross-attention launches) the text branch sits on the critical path and split-K is worth 2.7 %
# of the clip (9.24 -> 8.99 ms on one box, A/B in one gpurun call).  Stand-alone the 32x768x3072 projection drops from
# 43 us to ~12 us.
SPLITK_ENABLED = os.environ.get("TCE_SPLITK", "1") == "1"


def splitk_for(M, N, K):
    """Split count for skinny deep GEMMs (0/1 = do not split): fill ~256 workgroups, keep >= 128 of K per chunk."""
    blocks = ((M + 63) // 64) * ((N + 63) // 64)
    if not SPLITK_ENABLED:
        return 1
    if M > 512 or blocks >= 96 or K < 512 or N % 4:  # (up to 512 rows: a clip group's decoder / frame-token FFNs, the stride-32
        return 1                                      # level's 300 rows: 36 -> 12 us at 160-320 x 256 x 2048, tools/fewrow_bench.py rows)
    s = 1
    while s < 16 and blocks * s * 2 <= 384 and K % (s * 2 * 32) == 0 and K // (s * 2) >= 128:
        s *= 2
    return s


# ---------------------------------------------------------------------------------------------------------------
# Thin linear layers as weight streams (csrc/thin.hip): a few dozen rows against a wide / deep weight (RoBERTa at 32 tokens)
# ---------------------------------------------------------------------------------------------------------------
THIN_ENABLED = os.environ.get("TCE_THIN", "1") != "0"


def thin_splits(M, N, K):
    """Partial planes tce_thin_partials_f32 leaves for this shape (K / 256), or 0 when the shape / arithmetic is not its."""
    if not THIN_ENABLED or get_gemm_mode() == "f32":
        return 0
    s = lib().tce_thin_linear_splits(M, N, K)
    return s if s > 0 else 0


def thin_partials(x, w, ws, M, N, K, ldx=None, xsplits=0, bias_x=None, act_x=ACT_NONE):
    """ws[s][M][N] (s < K/256) = partial sums of x W^T.  xsplits > 0: x is itself [xsplits][M][K] partial planes of the previous
    layer, finished on load as act_x(sum + bias_x)."""
    check(lib().tce_thin_partials_f32(x.data_ptr(), K if ldx is None else ldx, xsplits, bias_x.data_ptr() if bias_x is not None else None,
                                      act_x, w.data_ptr(), w.stride(0), ws.data_ptr(), M, N, K, _stream()), "tce_thin_partials_f32")
    return ws


def splitk_reduce(ws, splits, M, N, out, bias=None, act=ACT_NONE, res=None, ldres=0, res_mode=RES_NONE, ln=None, eps=1e-5, ldc=None):
    """out = LN?( epi( sum of the partial planes + bias ) )"""
    check(lib().tce_splitk_reduce_f32(ws.data_ptr(), splits, M, N, bias.data_ptr() if bias is not None else None, act,
                                      res.data_ptr() if res_mode != RES_NONE else None, ldres, res_mode, out.data_ptr(),
                                      N if ldc is None else ldc, ln[0].data_ptr() if ln is not None else None,
                                      ln[1].data_ptr() if ln is not None else None, eps, _stream()), "tce_splitk_reduce_f32")
    return out


def select_masks(pred_logits, pred_masks, out_hw, threshold=0.5):
    """Caller harness H (inference_ytvos.py:238-250) on the GPU: pred_logits [T,Q,K], pred_masks [T,Q,h,w]
    -> (uint8 masks [T,H0,W0], best-query index tensor [1] int32)."""
    _chk(pred_logits, "pred_logits")
    _chk(pred_masks, "pred_masks")
    T, Q, K = pred_logits.shape
    _, _, h, w = pred_masks.shape
    H0, W0 = int(out_hw[0]), int(out_hw[1])
    out = torch.empty(T, H0, W0, dtype=torch.uint8, device=pred_masks.device)
    best = torch.empty(1, dtype=torch.int32, device=pred_masks.device)
    check(lib().tce_select_masks_u8(pred_logits.contiguous().data_ptr(), pred_masks.contiguous().data_ptr(),
                                    out.data_ptr(), best.data_ptr(), T, Q, K, h, w, H0, W0, float(threshold), _stream()),
          "tce_select_masks_u8")
    return out, best


def ffn_pack(w1, b1, w2, out=None):
    """Packs nn.Linear weights W1 [Hd,C], b1 [Hd], W2 [C,Hd] into the fused-FFN stream (csrc/chain.hip): fp16 hi/lo
    planes in MFMA-fragment order.  Done once per load_state_dict (static weights) or once per clip into an arena
    buffer `out` (the text cross-attention's folded weights)."""
    _chk(w1, "w1")
    _chk(w2, "w2")
    Hd, Cn = w1.shape
    if tuple(w2.shape) != (Cn, Hd):
        raise ValueError("ffn_pack: W2 must be [C, Hd]")
    nbytes = lib().tce_ffn_packed_bytes(Cn, Hd)
    if nbytes < 0:
        raise ValueError(f"ffn_pack: unsupported shape C={Cn} hidden={Hd}")
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=w1.device)
    elif out.numel() < nbytes:
        raise ValueError("ffn_pack: out buffer too small")
    check(lib().tce_ffn_pack_f32(w1.contiguous().data_ptr(), b1.contiguous().data_ptr() if b1 is not None else None,
                                 w2.contiguous().data_ptr(), out.data_ptr(), Cn, Hd, _stream()), "tce_ffn_pack_f32")
    return out


def ffn_supported(Cn, Hd):
    return lib().tce_ffn_packed_bytes(int(Cn), int(Hd)) > 0


FFN_SPLIT = os.environ.get("TCE_FFN_SPLIT", "1") != "0"  # A/B: 0 = a row block's hidden extent is never cut


def ffn_split_need(M, Cn, Hd, act):
    """(workspace floats, counters) of the hidden-extent split planned for this launch shape, (0, 0) when none is."""
    if not FFN_SPLIT:
        return 0, 0
    return int(lib().tce_ffn_split_ws_floats(M, Cn, Hd, act)), int(lib().tce_ffn_split_counters(M, Cn, Hd, act))


def ffn_fused(x, packed, b2, Hd, act, ln_in=None, ln_out=None, eps_in=1e-5, eps_out=1e-5, out=None, M=None, split=None):
    """out = LN_out?(x + W2 act(W1 LN_in?(x) + b1) + b2); x [M, C] (row pitch = x.stride(0)); ln_* = (gamma, beta).
    split = (workspace, counters) sized by ffn_split_need: the launch with the hidden extent cut (tce_ffn_fused_split_f32)."""
    _chk(x, "x")
    Cn = x.shape[-1]
    if M is None:
        M = x.numel() // Cn
    ldx = x.stride(0) if x.dim() == 2 else Cn
    if out is None:
        out = x
    ldo = out.stride(0) if out.dim() == 2 else Cn
    gi, bi = (ln_in[0].data_ptr(), ln_in[1].data_ptr()) if ln_in is not None else (None, None)
    go, bo = (ln_out[0].data_ptr(), ln_out[1].data_ptr()) if ln_out is not None else (None, None)
    def go_():
        if split is not None:
            ws, cnt = split
            assert cnt.dtype == torch.int32 and ws.dtype == torch.float32
            check(lib().tce_ffn_fused_split_f32(x.data_ptr(), ldx, packed.data_ptr(), b2.data_ptr(), gi, bi, eps_in, go, bo, eps_out,
                                                out.data_ptr(), ldo, M, Cn, Hd, act, ws.data_ptr(), ws.numel(), cnt.data_ptr(),
                                                cnt.numel(), _stream()), "tce_ffn_fused_split_f32")
            return
        check(lib().tce_ffn_fused_f32(x.data_ptr(), ldx, packed.data_ptr(), b2.data_ptr(), gi, bi, eps_in, go, bo, eps_out,
                                      out.data_ptr(), ldo, M, Cn, Hd, act, _stream()), "tce_ffn_fused_f32")
    if GEMM_PROFILE is None:
        go_()
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go_()
    e1.record()
    GEMM_PROFILE.append((f"ffn_fused_kernel<{Cn}", False, 4.0 * M * Cn * Hd, e0, e1))
    return out


# ---------------------------------------------------------------------------------------------------------------
# Token-stationary linear layers (csrc/chain.hip: tce_rowlin_f32) and the pixel-stationary 3x3 convolution.  Weights
# are packed once; a MODEL owns the packed copies of its weights in a Routes object (keyed by address + shape + the
# arithmetic they were packed in) and activates it for the duration of its launch program (`with ops.routes(r):`,
# pipeline.run_clip); gemm_ex / conv2d_cl route eligible launches through the ACTIVE table only.  Nothing is shared
# between models (a process-global table let one model's re-pack free streams that another model's captured
# graphs still pointed to), an entry lives exactly as long as the model's packed weights, and outside a `routes` block
# (direct ops calls with unrelated tensors) nothing is routed unless the caller registered it in the default table.
# ---------------------------------------------------------------------------------------------------------------
class Routes:
    def __init__(self):
        self.rowlin = {}   # (data_ptr, N, K, ldw, mode) of a [N, K] weight -> packed stream
        self.conv3 = {}    # (data_ptr, N, 9*Cin, mode) of a [N, 9*Cin] weight -> packed stream
        self.keep = []     # the source weights (views): their addresses are the keys, so they must outlive the entries

    def clear(self):
        self.rowlin.clear()
        self.conv3.clear()
        del self.keep[:]


DEFAULT_ROUTES = Routes()  # tools / tests that call the ops directly
_ROUTES = DEFAULT_ROUTES


class routes:
    """Activates a model's Routes for the launches inside the block."""

    def __init__(self, r):
        self.r, self.prev = r, None

    def __enter__(self):
        global _ROUTES
        self.prev, _ROUTES = _ROUTES, (self.r if self.r is not None else DEFAULT_ROUTES)
        return self

    def __exit__(self, *exc):
        global _ROUTES
        _ROUTES = self.prev
        return False


def current_routes():
    return _ROUTES


ROWLIN_MIN_ROWS = int(os.environ.get("TCE_ROWLIN_MIN_ROWS", 12000))
ROWLIN384_MAX_ROWS = int(os.environ.get("TCE_ROWLIN384_MAX_ROWS", 12000))
ROWLIN_K = (96, 128, 192, 256) + ((384,) if os.environ.get("TCE_ROWLIN_K384", "1") != "0" else ()) \
    + ((512,) if os.environ.get("TCE_ROWLIN_K512", "1") != "0" else ())  # A/B switches


def rowlin_pack(w, N=None, K=None, ldw=None):
    _chk(w, "w")
    N = w.shape[0] if N is None else N
    K = w.shape[1] if K is None else K
    ldw = w.stride(0) if ldw is None else ldw
    nbytes = lib().tce_rowlin_packed_bytes(N, K)
    if nbytes < 0:
        raise ValueError(f"rowlin_pack: unsupported shape N={N} K={K}")
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    check(lib().tce_rowlin_pack_f32(w.data_ptr(), ldw, out.data_ptr(), N, K, _stream()), "tce_rowlin_pack_f32")
    return out


def rowlin_register(w, table=None):
    """Packs a weight [N, K] (row-strided views welcome) in the CURRENT arithmetic and registers it in `table` (default: the
    active Routes); no-op if ineligible."""
    if w.dim() != 2 or w.stride(1) != 1 or w.shape[1] not in ROWLIN_K or w.shape[0] % 32 or not w.is_cuda:
        return None
    r = _ROUTES if table is None else table
    key = (w.data_ptr(), w.shape[0], w.shape[1], w.stride(0), get_gemm_mode())
    if key not in r.rowlin:
        r.rowlin[key] = rowlin_pack(w)
        r.keep.append(w)
    return r.rowlin[key]


def rowlin(x, pk, out, M, N, K, ldx, ldo, bias=None, a2=None, lda2=0, a2_rows=0, act=ACT_NONE, res=None, ldres=0,
           res_mode=RES_NONE, ln_in=None, ln_out=None, eps_in=1e-5, eps_out=1e-5, batch=1, sX=0, sA2=0, sRes=0, sOut=0):
    from ._lib import RowLinArgs
    q = RowLinArgs()
    q.x, q.packed, q.out = x.data_ptr(), pk.data_ptr(), out.data_ptr()
    q.M, q.N, q.K, q.batch = M, N, K, batch
    q.ldx, q.ldo = ldx, ldo
    if a2 is not None:
        q.a2, q.lda2, q.a2_rows = a2.data_ptr(), lda2, a2_rows
    if bias is not None:
        q.bias = bias.data_ptr()
    if res_mode != RES_NONE:
        q.res, q.ldres = res.data_ptr(), ldres
    if ln_in is not None:
        q.g_in, q.be_in = ln_in[0].data_ptr(), ln_in[1].data_ptr()
    if ln_out is not None:
        q.g_out, q.be_out = ln_out[0].data_ptr(), ln_out[1].data_ptr()
    q.act, q.res_mode, q.eps_in, q.eps_out = act, res_mode, eps_in, eps_out
    q.sX, q.sA2, q.sRes, q.sOut = sX, sA2, sRes, sOut

    def go():
        check(lib().tce_rowlin_f32(C.byref(q), _stream()), "tce_rowlin_f32")
    if GEMM_PROFILE is None:
        go()
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    GEMM_PROFILE.append((f"rowlin_kernel<{K}", False, 2.0 * M * N * K * batch, e0, e1))
    return out


# ---------------------------------------------------------------------------------------------------------------
# 3x3 / stride 1 / pad 1 convolution 256 -> 256 as a pixel-stationary launch (csrc/chain.hip: tce_conv3x3_f32)
# ---------------------------------------------------------------------------------------------------------------
CONV3_MIN_PIXELS = int(os.environ.get("TCE_CONV3_MIN_PIXELS", 12000))


def conv3x3_pack(w_cl, Cin):
    """w_cl [N, 9*Cin], k = (ky*3+kx)*Cin + c."""
    _chk(w_cl, "w")
    N = w_cl.shape[0]
    nbytes = lib().tce_conv3x3_packed_bytes(Cin, N)
    if nbytes < 0 or w_cl.shape[1] != 9 * Cin or not w_cl.is_contiguous():
        raise ValueError(f"conv3x3_pack: unsupported shape N={N} Cin={Cin}")
    out = torch.empty(nbytes, dtype=torch.uint8, device=w_cl.device)
    check(lib().tce_conv3x3_pack_f32(w_cl.data_ptr(), out.data_ptr(), Cin, N, _stream()), "tce_conv3x3_pack_f32")
    return out


def conv3x3_register(w_cl, Cin, table=None):
    if lib().tce_conv3x3_packed_bytes(Cin, w_cl.shape[0]) < 0 or w_cl.shape[1] != 9 * Cin:
        return None
    r = _ROUTES if table is None else table
    key = (w_cl.data_ptr(), w_cl.shape[0], w_cl.shape[1], get_gemm_mode())
    if key not in r.conv3:
        r.conv3[key] = conv3x3_pack(w_cl, Cin)
        r.keep.append(w_cl)
    return r.conv3[key]


def conv3x3(x, pk, T, H, W, Cin, N, bias=None, out=None, alloc=None):
    """x [T*H*W, Cin] channels-last -> [T*H*W, N]."""
    _chk(x, "x")
    M = T * H * W
    if out is None:
        out = alloc(M, N) if alloc else torch.empty(M, N, dtype=torch.float32, device=x.device)

    def go():
        check(lib().tce_conv3x3_f32(x.data_ptr(), x.stride(0), pk.data_ptr(), bias.data_ptr() if bias is not None else None,
                                    out.data_ptr(), out.stride(0), T, H, W, Cin, N, _stream()), "tce_conv3x3_f32")
    if GEMM_PROFILE is None:
        go()
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    GEMM_PROFILE.append((f"conv3x3_kernel<{Cin}", True, 2.0 * M * N * 9 * Cin, e0, e1))
    return out


# ---------------------------------------------------------------------------------------------------------------
# Text cross-attention as one token-stationary launch (csrc/chain.hip: tce_xattn_prepare_f32 / tce_xattn_fused_f32):
# with <= 32 keys the per-head scores are linear in the query rows, so q-projection -> attention -> out-projection ->
# residual -> LayerNorm is the fused FFN kernel with a grouped softmax as its activation.
# ---------------------------------------------------------------------------------------------------------------
XATTN_MIN_ROWS = int(os.environ.get("TCE_XATTN_MIN_ROWS", 4000))
XATTN_PACK_FUSED = os.environ.get("TCE_XATTN_PACK_FUSED", "1") != "0"


def xattn_static(wq, bq, scale=32 ** -0.5):
    """Static half of the fold, done at pack time: scale * [W_q^T ; b_q] as a contiguous [257, 256] tensor."""
    return (torch.cat([wq.t(), bq[None]], 0) * scale).contiguous()


def xattn_pack(k, v, wqT_ext, wo, L, alloc, group=32, batch=1):
    """Per clip: folds the projected keys / values ([batch][L,256]) of the 8 heads into W1, b1, W2 and packs the weight
    stream(s).  Returns a uint8 tensor [batch, bytes_per_stream]."""
    _chk(k, "k")
    _chk(v, "v")
    Hd = 8 * group
    if XATTN_PACK_FUSED:  # fold + pack in one launch (bit-identical stream)
        pk = alloc(batch, lib().tce_ffn_packed_bytes(256, Hd), dtype=torch.uint8)
        check(lib().tce_xattn_pack_f32(k.data_ptr(), v.data_ptr(), wqT_ext.data_ptr(), wo.data_ptr(), pk.data_ptr(), L, group, batch,
                                       _stream()), "tce_xattn_pack_f32")
        return pk
    W1, b1, W2 = alloc(batch, Hd, 256), alloc(batch, Hd), alloc(batch, 256, Hd)
    check(lib().tce_xattn_prepare_f32(k.data_ptr(), v.data_ptr(), wqT_ext.data_ptr(), wo.data_ptr(), W1.data_ptr(), b1.data_ptr(),
                                      W2.data_ptr(), L, group, batch, _stream()), "tce_xattn_prepare_f32")
    nbytes = lib().tce_ffn_packed_bytes(256, Hd)
    pk = alloc(batch, nbytes, dtype=torch.uint8)
    check(lib().tce_ffn_pack_batched_f32(W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), pk.data_ptr(), 256, Hd, batch, _stream()),
          "tce_ffn_pack_batched_f32")
    return pk


# cross-attention and the FFN behind it as ONE launch (tce_xattn_ffn_fused_f32).  Built and measured in round 5: correct (kernel and
# whole-clip parity green with it on) but not faster -- config 2 B=1 6.37 / 6.41 ms with it against 6.29 / 6.36 ms without, G=8 37.5
# against 36.8 ms (profiles/r05_round_ab.txt): the launch saved and the [M, 256] re-read avoided (~5 us a site) are less than what
# the second stage loses to the chain kernel's register pressure (50 spilled VGPRs outside the loops, one resident workgroup).  Off.
XATTN_FFN_CHAIN = os.environ.get("TCE_XATTN_FFN_CHAIN", "0") != "0"


def ffn_pack_chain(w1, b1, w2):
    """FFN stream for the SECOND stage of a cross-attention -> FFN chain launch (tce_ffn_pack_chain_f32: W1 in the k order of the
    accumulator registers; C = 256)."""
    _chk(w1, "w1")
    _chk(w2, "w2")
    Hd, Cn = w1.shape
    nbytes = lib().tce_ffn_packed_bytes(Cn, Hd)
    if Cn != 256 or nbytes < 0 or tuple(w2.shape) != (Cn, Hd):
        raise ValueError(f"ffn_pack_chain: unsupported shape C={Cn} hidden={Hd}")
    out = torch.empty(nbytes, dtype=torch.uint8, device=w1.device)
    check(lib().tce_ffn_pack_chain_f32(w1.contiguous().data_ptr(), b1.contiguous().data_ptr() if b1 is not None else None,
                                       w2.contiguous().data_ptr(), out.data_ptr(), Cn, Hd, _stream()), "tce_ffn_pack_chain_f32")
    return out


def xattn_fused(x, pk, bo, M, out, a2=None, lda2=256, a2_rows=0, res=None, res_mode=RES_ADD, ln_out=None, eps_out=1e-5,
                batch=1, sX=0, sRes=0, sOut=0, ldx=256, ldo=256, ldres=256, group=32, per_batch_weights=False, w_div=1, ffn=None):
    """w_div: batch entries that share one weight stream (entry b reads stream b // w_div; per_batch_weights only).
    ffn = (chain-packed stream, b2, hidden, (gamma, beta) or None, mid tensor, sMid): the FFN that follows runs in the SAME launch
    (tce_xattn_ffn_fused_f32): `out` receives LN(y + FFN(y)), `mid` [rows, 256] the attention stage's y."""
    from ._lib import XattnArgs, XattnFfnArgs
    q = XattnArgs()
    q.x, q.packed, q.bo, q.out = x.data_ptr(), pk.data_ptr(), bo.data_ptr(), out.data_ptr()
    q.M, q.batch, q.res_mode, q.eps_out, q.group = M, batch, res_mode, eps_out, group
    q.sW = pk.stride(0) if (per_batch_weights and pk.dim() == 2) else 0
    q.w_div = max(1, int(w_div))
    q.ldx, q.ldo, q.ldres, q.lda2 = ldx, ldo, ldres, lda2
    q.sX, q.sRes, q.sOut = sX, sRes, sOut
    if a2 is not None:
        q.a2, q.a2_rows = a2.data_ptr(), a2_rows
    if res is not None:
        q.res = res.data_ptr()
    if ln_out is not None:
        q.g_out, q.be_out = ln_out[0].data_ptr(), ln_out[1].data_ptr()

    fq = None
    if ffn is not None:
        pk2, b2, hidden, ln2, mid, s_mid = ffn
        fq = XattnFfnArgs()
        fq.packed, fq.b2, fq.mid, fq.ldmid, fq.sMid, fq.hidden, fq.act = pk2.data_ptr(), b2.data_ptr(), mid.data_ptr(), 256, s_mid, hidden, 1
        if ln2 is not None:
            fq.g_out, fq.be_out, fq.eps_out = ln2[0].data_ptr(), ln2[1].data_ptr(), 1e-5

    def go():
        if fq is not None:
            check(lib().tce_xattn_ffn_fused_f32(C.byref(q), C.byref(fq), _stream()), "tce_xattn_ffn_fused_f32")
        else:
            check(lib().tce_xattn_fused_f32(C.byref(q), _stream()), "tce_xattn_fused_f32")
    if GEMM_PROFILE is None:
        go()
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    if fq is not None:
        GEMM_PROFILE.append(("xattn+ffn(ffn_fused_kernel<256,4,3|4,1>", False, 4.0 * M * 256 * (8 * group + hidden) * batch, e0, e1))
    else:
        GEMM_PROFILE.append(("xattn(ffn_fused_kernel<256,4,3|4>", False, 4.0 * M * 256 * 8 * group * batch, e0, e1))
    return out


# ---------------------------------------------------------------------------------------------------------------
# Few-row linear layers (csrc/fewrow.hip: tce_fewrow_linear_f32): the per-token projections of the frame-token path, the
# decoder's per-query projections, the text-side key / value projections -- a few dozen rows, exact fp32, up to three
# projections of the same rows per launch.
# ---------------------------------------------------------------------------------------------------------------
# Routed by ROWS: the kernel re-stages its <= 32 x K row block once per 8 output columns, so above ~128 rows it does
# GEMM-sized work on the VALU (clip groups: 320 token rows, 800 controller rows at G = 8 -- 56 us per launch on average in
# profiles/r04_kernel_stats_cfg2_group8.csv); those sites take the tiled / split-K GEMM path again.
FEWROW_MAX_ROWS = int(os.environ.get("TCE_FEWROW_MAX_ROWS", 128))
FR_NONE, FR_RELU, FR_SIGMOID, FR_GELU = 0, 1, 2, 3


def fewrow_linear(x, R, K, segs, ldx=None, a2=None, lda2=0, a2_rows=0, res=None, ldres=0, ln_in=None, eps_in=1e-5, xn_out=None):
    """segs: list of (W [N,K] (row-strided view ok), bias or None, out tensor, N, ldo, use_a2, act); res: added to segment 0.
    ln_in = (gamma, beta): LayerNorm of the x rows BEFORE the addend and the projections (K = 256); xn_out: the normalised rows
    are also written there (a tensor that overlaps none of the operands)."""
    from ._lib import FewRowArgs
    q = FewRowArgs()
    if ln_in is not None:
        q.g_in, q.be_in, q.eps_in = ln_in[0].data_ptr(), ln_in[1].data_ptr(), eps_in
        if xn_out is not None:
            q.xn_out, q.ldxn = xn_out.data_ptr(), (xn_out.stride(0) if xn_out.dim() == 2 else K)
    q.x, q.ldx, q.R, q.K, q.nseg = x.data_ptr(), (K if ldx is None else ldx), R, K, len(segs)
    if a2 is not None:
        q.a2, q.lda2, q.a2_rows = a2.data_ptr(), lda2, a2_rows
    if res is not None:
        q.res, q.ldres = res.data_ptr(), ldres
    for i, (W, b, out, N, ldo, use_a2, act) in enumerate(segs):
        sg = q.seg[i]
        sg.W, sg.bias, sg.out = W.data_ptr(), (b.data_ptr() if b is not None else None), out.data_ptr()
        sg.N, sg.ldw, sg.ldo, sg.use_a2, sg.act = N, (W.stride(0) if W.dim() == 2 else K), ldo, 1 if use_a2 else 0, act
    check(lib().tce_fewrow_linear_f32(C.byref(q), _stream()), "tce_fewrow_linear_f32")
