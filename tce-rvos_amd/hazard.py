"""Launch-program hazard checker: proves (or refutes) that the clip's multi-stream launch program is race free.

The per-clip forward is ONE launch program issued on up to six streams (pipeline._Fork); captured, every fork / join
becomes a graph edge and everything else runs concurrently.  A missing edge, an arena range handed out again by
`release()` while another branch still reads it, or two branches sharing a workspace are invisible to the parity tests
when the timing happens to hide them -- and a captured graph replays that luck forever (a replay that
differed from the eager pass).  This module needs no failure to come back: it RECORDS one pass of the real program and
checks it statically.

  * every C-ABI launch (`lib().tce_*`) is intercepted: the stream it was issued on and -- from the same pointers and sizes
    the kernel receives -- the exact byte ranges it reads and writes (`MODELS`, one access model per entry point of
    include/tce_rvos.h; an entry point without a model is an error, so coverage cannot rot);
  * every event record / wait (torch's `wait_stream` is `wait_event(record_event())`) and every host synchronisation is
    intercepted and turned into vector clocks: launch A happens-before launch B iff B's clock has seen A's tick;
  * `analyse()` then asserts that any two launches NOT ordered by happens-before touch disjoint memory (write/write and
    write/read), which covers arena reuse by construction (addresses, not names, are compared).

It observes; it computes nothing and changes no result.  model.hazard_check(...) runs it on the capture topology.
"""
import contextlib
import traceback

import numpy as np
import torch

from . import _lib

F = 4  # sizeof(float)


# ---------------------------------------------------------------------------------------------------------------------
# byte-range sets
# ---------------------------------------------------------------------------------------------------------------------
def strided(ptr, run_bytes, *dims):
    """Intervals [start, end) of `run_bytes` contiguous bytes repeated over dims = (count, stride_bytes), outermost first.
    Returns an int64 array [n, 2], sorted and merged; empty for a NULL pointer or an empty extent."""
    if not ptr or run_bytes <= 0 or any(c <= 0 for c, _ in dims):
        return np.zeros((0, 2), dtype=np.int64)
    starts = np.array([int(ptr)], dtype=np.int64)
    for count, stride in dims:
        if count == 1 or stride == 0:
            continue
        starts = (starts[:, None] + (np.arange(int(count), dtype=np.int64) * int(stride))[None, :]).reshape(-1)
    iv = np.stack([starts, starts + int(run_bytes)], 1)
    return merge(iv)


def dense(ptr, nbytes):
    return strided(ptr, int(nbytes))


def merge(iv):
    if len(iv) <= 1:
        return iv.reshape(-1, 2)
    iv = iv[np.argsort(iv[:, 0], kind="stable")]
    end = np.maximum.accumulate(iv[:, 1])
    new = np.ones(len(iv), dtype=bool)
    new[1:] = iv[1:, 0] > end[:-1]  # touching intervals merge too
    idx = np.flatnonzero(new)
    return np.stack([iv[idx, 0], np.concatenate([end[idx[1:] - 1], end[-1:]])], 1)


def union(*sets):
    sets = [s for s in sets if len(s)]
    if not sets:
        return np.zeros((0, 2), dtype=np.int64)
    return merge(np.concatenate(sets, 0))


def overlap(a, b):
    """First overlapping byte range of two merged interval sets, or None."""
    if not len(a) or not len(b) or a[0, 0] >= b[-1, 1] or b[0, 0] >= a[-1, 1]:
        return None
    if len(a) > len(b):
        a, b = b, a
    # for every interval of a: the first interval of b that ends after a's start
    j = np.searchsorted(b[:, 1], a[:, 0], side="right")
    ok = j < len(b)
    hit = np.zeros(len(a), dtype=bool)
    hit[ok] = b[j[ok], 0] < a[ok, 1]
    k = np.flatnonzero(hit)
    if not len(k):
        return None
    i = int(k[0])
    return max(int(a[i, 0]), int(b[j[i], 0])), min(int(a[i, 1]), int(b[j[i], 1]))


# ---------------------------------------------------------------------------------------------------------------------
# access models: name -> fn(args) -> (reads, writes) as lists of interval sets.  `a` are the Python-side arguments of the
# ctypes call (ints / None for pointers, byref(struct) for argument blocks), a[-1] is the stream.
# ---------------------------------------------------------------------------------------------------------------------
def _st(x):
    """The ctypes structure behind byref(struct) / a pointer / the structure itself."""
    if hasattr(x, "_obj"):
        return x._obj
    if hasattr(x, "contents"):
        return x.contents
    return x


def _p(v):
    return int(v) if v else 0


def _gemm(g, ws=None, splits=1, ln=None):
    b = max(1, g.batch)
    if g.conv:
        rd = [dense(_p(g.A), g.T * g.H * g.Wd * g.Cin * F)]
    else:
        rd = [strided(_p(g.A), g.K * F, (b, g.sA * F), (g.M, g.lda * F))]
        if g.A2:
            rd.append(strided(_p(g.A2), g.K * F, (b, g.sA2 * F), (g.M, g.lda2 * F)))
    rd.append(strided(_p(g.W), g.K * F, (b, g.sW * F), (g.N, g.ldw * F)))
    if g.bias:
        rd.append(strided(_p(g.bias), g.N * F, (b, g.sBias * F)))
    if g.res and g.res_mode:
        rd.append(strided(_p(g.res), g.N * F, (b, g.sRes * F), (g.M, g.ldres * F)))
    wr = [strided(_p(g.C), g.N * F, (b, g.sC * F), (g.M, g.ldc * F))]
    if ws:
        w = dense(_p(ws), splits * g.M * g.N * F)
        rd.append(w)
        wr.append(w)
    if ln:
        rd += [dense(_p(ln[0]), g.N * F), dense(_p(ln[1]), g.N * F)]
    return rd, wr


def _rowlin(q):
    b = max(1, q.batch)
    L = _lib.lib_raw()
    rd = [strided(_p(q.x), q.K * F, (b, q.sX * F), (q.M, q.ldx * F)),
          dense(_p(q.packed), L.tce_rowlin_packed_bytes(q.N, q.K))]
    if q.a2:
        rd.append(strided(_p(q.a2), q.K * F, (b, q.sA2 * F), (q.a2_rows if q.a2_rows > 0 else q.M, q.lda2 * F)))
    if q.bias:
        rd.append(dense(_p(q.bias), q.N * F))
    if q.res and q.res_mode:
        rd.append(strided(_p(q.res), q.N * F, (b, q.sRes * F), (q.M, q.ldres * F)))
    for g_, n in ((q.g_in, q.K), (q.be_in, q.K), (q.g_out, q.N), (q.be_out, q.N)):
        if g_:
            rd.append(dense(_p(g_), n * F))
    return rd, [strided(_p(q.out), q.N * F, (b, q.sOut * F), (q.M, q.ldo * F))]


def _xattn(q):
    b = max(1, q.batch)
    L = _lib.lib_raw()
    nb = L.tce_ffn_packed_bytes(256, 8 * q.group)
    wd = q.w_div if q.w_div > 0 else 1
    rd = [strided(_p(q.x), 256 * F, (b, q.sX * F), (q.M, q.ldx * F)),
          strided(_p(q.packed), nb, ((b + wd - 1) // wd if q.sW else 1, q.sW)), dense(_p(q.bo), 256 * F)]
    if q.a2:
        rd.append(strided(_p(q.a2), 256 * F, (q.a2_rows if q.a2_rows > 0 else q.M, q.lda2 * F)))
    if q.res:
        rd.append(strided(_p(q.res), 256 * F, (b, q.sRes * F), (q.M, q.ldres * F)))
    for g_ in (q.g_out, q.be_out):
        if g_:
            rd.append(dense(_p(g_), 256 * F))
    return rd, [strided(_p(q.out), 256 * F, (b, q.sOut * F), (q.M, q.ldo * F))]


def _xattn_ffn(q, f):
    """the chain launch: the attention stage's accesses + the FFN stage's stream / bias / LayerNorm; `mid` is written and re-read"""
    rd, wr = _xattn(q)
    b = max(1, q.batch)
    rd += [dense(_p(f.packed), _lib.lib_raw().tce_ffn_packed_bytes(256, f.hidden)), dense(_p(f.b2), 256 * F)]
    for g_ in (f.g_out, f.be_out):
        if g_:
            rd.append(dense(_p(g_), 256 * F))
    wr.append(strided(_p(f.mid), 256 * F, (b, f.sMid * F), (q.M, f.ldmid * F)))
    return rd, wr


def _fewrow(q):
    rd = [strided(_p(q.x), q.K * F, (q.R, q.ldx * F))]
    wr = []
    use_a2 = False
    for i in range(q.nseg):
        sg = q.seg[i]
        rd.append(strided(_p(sg.W), q.K * F, (sg.N, sg.ldw * F)))
        if sg.bias:
            rd.append(dense(_p(sg.bias), sg.N * F))
        wr.append(strided(_p(sg.out), sg.N * F, (q.R, sg.ldo * F)))
        use_a2 = use_a2 or bool(sg.use_a2)
    if q.a2 and use_a2:
        rd.append(strided(_p(q.a2), q.K * F, (q.a2_rows if q.a2_rows > 0 else q.R, q.lda2 * F)))
    if q.res:
        rd.append(strided(_p(q.res), q.seg[0].N * F, (q.R, q.ldres * F)))
    if q.g_in:  # LayerNorm prologue (round 5)
        rd += [dense(_p(q.g_in), q.K * F), dense(_p(q.be_in), q.K * F)]
        if q.xn_out:
            wr.append(strided(_p(q.xn_out), q.K * F, (q.R, q.ldxn * F)))
    return rd, wr


def _mha(Q, K, V, O, ws, batch, nh, Lq, Lk, ldq, ldk, ldv, ldo, sQ, sK, sV, sO, kmask):
    run = nh * 32 * F
    rd = [strided(_p(Q), run, (batch, sQ * F), (Lq, ldq * F)), strided(_p(K), run, (batch, sK * F), (Lk, ldk * F)),
          strided(_p(V), run, (batch, sV * F), (Lk, ldv * F))]
    if kmask:
        rd.append(dense(_p(kmask), batch * Lk))
    wr = [strided(_p(O), run, (batch, sO * F), (Lq, ldo * F))]
    if ws:
        w = dense(_p(ws), _lib.lib_raw().tce_mha_ws_bytes(batch, nh, Lk))
        rd.append(w)
        wr.append(w)
    return rd, wr


def _msda(value, proj, ref, out, N, S, M, Lq, L, P, ref_dim, ref_per_frame):
    return ([dense(_p(value), N * S * M * 32 * F), dense(_p(proj), N * Lq * M * L * P * 3 * F),
             dense(_p(ref), (N if ref_per_frame else 1) * Lq * ref_dim * F)], [dense(_p(out), N * Lq * M * 32 * F)])


def _copy_segments(a):
    segs, n = a[0], a[1]
    rd, wr = [], []
    for i in range(n):
        sg = segs[i]
        rd.append(strided(_p(sg.src), sg.row_words * 4, (sg.rows, sg.src_pitch_words * 4)))
        wr.append(dense(_p(sg.dst), sg.rows * sg.row_words * 4))
    return rd, wr


def _ffn_fused(a):
    x, ldx, packed, b2, g_in, be_in, _, g_out, be_out, _, out, ldo, M, Cn, Hd = a[:15]
    rd = [strided(_p(x), Cn * F, (M, ldx * F)), dense(_p(packed), _lib.lib_raw().tce_ffn_packed_bytes(Cn, Hd)),
          dense(_p(b2), Cn * F)]
    for g_ in (g_in, be_in, g_out, be_out):
        if g_:
            rd.append(dense(_p(g_), Cn * F))
    return rd, [strided(_p(out), Cn * F, (M, ldo * F))]


def _ffn_fused_split(a):
    rd, wr = _ffn_fused(a)
    ws, ws_floats, cnt, ncnt = a[16:20]
    # the workspace and the counters are written and read inside the launch
    return rd, wr + [dense(_p(ws), ws_floats * F), dense(_p(cnt), ncnt * 4)]


def _ffn_pack(a, batched):
    W1, b1, W2, packed, Cn, Hd = a[:6]
    b = a[6] if batched else 1
    nb = _lib.lib_raw().tce_ffn_packed_bytes(Cn, Hd)
    rd = [dense(_p(W1), b * Hd * Cn * F), dense(_p(W2), b * Hd * Cn * F)]
    if b1:
        rd.append(dense(_p(b1), b * Hd * F))
    return rd, [dense(_p(packed), b * nb)]


def _groupnorm(a):
    x, gamma, beta, out, ws, T, HW, Cn, G = a[:9]
    nsplit = _lib.lib_raw().tce_groupnorm_nsplit(HW)
    w = dense(_p(ws), T * G * (nsplit * 3 + 2) * F)
    return ([dense(_p(x), T * HW * Cn * F), dense(_p(gamma), Cn * F), dense(_p(beta), Cn * F), w],
            [dense(_p(out), T * HW * Cn * F), w])


def _groupnorm_up_add(a):
    x, gamma, beta, add, out, ws, T, h, w_, ho, wo, Cn, G = a[:13]
    nsplit = _lib.lib_raw().tce_groupnorm_nsplit(h * w_)
    w = dense(_p(ws), T * G * (nsplit * 3 + 2) * F)
    return ([dense(_p(x), T * h * w_ * Cn * F), dense(_p(gamma), Cn * F), dense(_p(beta), Cn * F), dense(_p(add), T * ho * wo * Cn * F), w],
            [dense(_p(out), T * ho * wo * Cn * F), w])


def _mask_pack(a):
    params, w0f, tail, nl, T, Q, Cm = a[:7]
    npar = 8 * (Cm + 2) + 64 + 8 + 8 + 8 + 1
    return [dense(_p(params), nl * T * Q * npar * F)], [dense(_p(w0f), T * nl * Q * 8 * Cm * F), dense(_p(tail), nl * T * Q * 112 * F)]


def _mask_tail(a):
    G, tail, refs, ref_ld, masks, nl, T, Q, h, w = a[:10]
    return ([dense(_p(G), T * h * w * nl * Q * 8 * F), dense(_p(tail), nl * T * Q * 112 * F), dense(_p(refs), nl * T * Q * ref_ld * F)],
            [dense(_p(masks), nl * T * Q * h * w * F)])


def _resize(a):
    src, add, out, T, h, w, ho, wo, Cn = a[:9]
    rd = [dense(_p(src), T * h * w * Cn * F)]
    if add:
        rd.append(dense(_p(add), T * ho * wo * Cn * F))
    return rd, [dense(_p(out), T * ho * wo * Cn * F)]


def _resize_ln(a):
    src, add, gamma, beta, _, out, T, h, w, ho, wo, Cn = a[:12]
    return ([dense(_p(src), T * h * w * Cn * F), dense(_p(add), T * ho * wo * Cn * F), dense(_p(gamma), Cn * F), dense(_p(beta), Cn * F)],
            [dense(_p(out), T * ho * wo * Cn * F)])


def _win(a):
    qkv, qb, table, out, T, H, W, Cn, nH = a[:9]
    return ([dense(_p(qkv), T * H * W * 3 * Cn * F), dense(_p(qb), 3 * Cn * F)], [dense(_p(out), T * H * W * Cn * F)])


def _embed(a):
    ids, pos_ids, word, pos, type0, gamma, beta, out, L, Cn = a[:10]
    rd = [dense(_p(ids), L * 8), dense(_p(gamma), Cn * F), dense(_p(beta), Cn * F)]
    if pos_ids:
        rd.append(dense(_p(pos_ids), L * 8))
    return rd, [dense(_p(out), L * Cn * F)]


def _pos(a, valid):
    out, add, T, h, w, Fh = a[:6]
    return ([dense(_p(add), 2 * Fh * F)] if add else []), [dense(_p(out), T * h * w * 2 * Fh * F)]


MODELS = {
    "tce_gemm_f32": lambda a: _gemm(_st(a[0])),
    "tce_gemm_splitk_f32": lambda a: _gemm(_st(a[0]), ws=a[2], splits=a[1]),
    "tce_gemm_splitk_ln_f32": lambda a: _gemm(_st(a[0]), ws=a[2], splits=a[1], ln=(a[3], a[4])),
    "tce_layernorm_f32": lambda a: ([dense(_p(a[0]), a[5] * a[6] * F), dense(_p(a[1]), a[5] * a[6] * F), dense(_p(a[2]), a[6] * F),
                                     dense(_p(a[3]), a[6] * F)], [dense(_p(a[4]), a[5] * a[6] * F)]),
    "tce_groupnorm_f32": _groupnorm,
    "tce_groupnorm_up_add_f32": _groupnorm_up_add,
    "tce_resnet_stem_f32": lambda a: ([dense(_p(a[0]), a[4] * 3 * a[5] * a[6] * F), dense(_p(a[1]), 147 * 64 * F), dense(_p(a[2]), 64 * F)],
                                      [dense(_p(a[3]), a[4] * ((a[5] - 1) // 2 + 1) * ((a[6] - 1) // 2 + 1) * 64 * F)]),
    "tce_maxpool3x3s2_cl_f32": lambda a: ([dense(_p(a[0]), a[2] * a[3] * a[4] * a[5] * F)],
                                          [dense(_p(a[1]), a[2] * ((a[3] - 1) // 2 + 1) * ((a[4] - 1) // 2 + 1) * a[5] * F)]),
    "tce_patch_embed_f32": lambda a: ([dense(_p(a[0]), a[6] * 3 * a[7] * a[8] * F), dense(_p(a[1]), a[9] * 48 * F)],
                                      [dense(_p(a[5]), a[6] * ((a[7] + 3) // 4) * ((a[8] + 3) // 4) * a[9] * F)]),
    "tce_window_attn_f32": _win,
    "tce_window_attn3d_f32": _win,
    "tce_patch_merge_ln_f32": lambda a: ([dense(_p(a[0]), a[4] * a[5] * a[6] * a[7] * F)],
                                         [dense(_p(a[3]), a[4] * ((a[5] + 1) // 2) * ((a[6] + 1) // 2) * 4 * a[7] * F)]),
    "tce_mha_f32": lambda a: _mha(a[0], a[1], a[2], a[3], None, *a[4:17]),
    "tce_mha_ws_f32": lambda a: _mha(a[0], a[1], a[2], a[3], a[4], *a[5:18]),
    "tce_ms_deform_attn_forward_f32": lambda a: (
        [dense(_p(a[0]), a[6] * a[7] * a[8] * a[9] * F), dense(_p(a[1]), a[11] * 16), dense(_p(a[2]), a[11] * 8),
         dense(_p(a[3]), a[6] * a[10] * a[8] * a[11] * a[12] * 2 * F), dense(_p(a[4]), a[6] * a[10] * a[8] * a[11] * a[12] * F)],
        [dense(_p(a[5]), a[6] * a[10] * a[8] * a[9] * F)]),
    "tce_ms_deform_attn_backward_f32": lambda a: (
        [dense(_p(a[0]), a[9] * a[10] * a[11] * a[12] * F), dense(_p(a[3]), a[9] * a[13] * a[11] * a[14] * a[15] * 2 * F),
         dense(_p(a[4]), a[9] * a[13] * a[11] * a[14] * a[15] * F), dense(_p(a[5]), a[9] * a[13] * a[11] * a[12] * F)],
        [dense(_p(a[6]), a[9] * a[10] * a[11] * a[12] * F), dense(_p(a[7]), a[9] * a[13] * a[11] * a[14] * a[15] * 2 * F),
         dense(_p(a[8]), a[9] * a[13] * a[11] * a[14] * a[15] * F)]),
    "tce_msda_fused_f32": lambda a: _msda(a[0], a[1], a[2], a[3], *a[5:13]),
    "tce_msda_fused_valid_f32": lambda a: _msda(a[0], a[1], a[2], a[3], *a[6:14]),
    # (src, wv, bv, proj, ref, out, shapes, valid, N, S, M, Lq, L, P, ref_dim, ref_per_frame): reads the un-projected rows
    "tce_msda_fewq_raw_f32": lambda a: (
        [dense(_p(a[0]), a[8] * a[9] * 256 * F), dense(_p(a[1]), 256 * 256 * F), dense(_p(a[2]), 256 * F),
         dense(_p(a[3]), a[8] * a[11] * a[10] * a[12] * a[13] * 3 * F), dense(_p(a[4]), (a[8] if a[15] else 1) * a[11] * a[14] * F)],
        [dense(_p(a[5]), a[8] * a[11] * 256 * F)]),
    # (memory, sent, out, ws, T, S, C, frames_per_clip)
    "tce_contrastive_f32": lambda a: ([dense(_p(a[0]), a[4] * a[5] * a[6] * F), dense(_p(a[1]), (a[4] // a[7]) * a[6] * F)],
                                      [dense(_p(a[2]), a[4] * F), dense(_p(a[3]), a[4] * 32 * a[6] * F)]),
    "tce_pos_sine2d_f32": lambda a: _pos(a, False),
    "tce_pos_sine2d_valid_f32": lambda a: _pos(a, True),
    "tce_resize_nearest_f32": _resize,
    "tce_resize_bilinear_f32": _resize,
    "tce_resize_bilinear_ln_f32": _resize_ln,
    "tce_add_f32": lambda a: ([dense(_p(a[0]), a[3] * F), dense(_p(a[1]), a[4] * F)], [dense(_p(a[2]), a[3] * F)]),
    "tce_tile_f32": lambda a: ([dense(_p(a[0]), a[2] * F)], [dense(_p(a[1]), a[2] * a[3] * F)]),
    "tce_sigmoid_f32": lambda a: ([dense(_p(a[0]), a[2] * F)], [dense(_p(a[1]), a[2] * F)]),
    "tce_tanh_f32": lambda a: ([dense(_p(a[0]), a[2] * F)], [dense(_p(a[1]), a[2] * F)]),
    "tce_copy_segments": _copy_segments,
    "tce_box_refine_f32": lambda a: ([dense(_p(a[0]), a[3] * 4 * F), dense(_p(a[1]), a[3] * a[4] * F)], [dense(_p(a[2]), a[3] * 4 * F)]),
    "tce_mask_pack_f32": _mask_pack,
    "tce_mask_tail_f32": _mask_tail,
    "tce_select_masks_u8": lambda a: ([dense(_p(a[0]), a[4] * a[5] * a[6] * F), dense(_p(a[1]), a[4] * a[5] * a[7] * a[8] * F)],
                                      [dense(_p(a[2]), a[4] * a[9] * a[10]), dense(_p(a[3]), 4)]),
    "tce_resize_h_u8": lambda a: ([dense(_p(a[0]), a[4] * a[5] * 3)], [dense(_p(a[3]), a[4] * a[6] * 3)]),
    "tce_resize_v_norm_f32": lambda a: ([dense(_p(a[0]), a[5] * a[6] * a[7] * 3)], [dense(_p(a[4]), a[5] * 3 * a[8] * a[7] * F)]),
    "tce_embed_ln_f32": _embed,
    "tce_mha_small64_f32": lambda a: ([dense(_p(a[0]), a[2] * 3 * a[3] * 64 * F)], [dense(_p(a[1]), a[2] * a[3] * 64 * F)]),
    "tce_ffn_pack_f32": lambda a: _ffn_pack(a, False),
    "tce_ffn_pack_batched_f32": lambda a: _ffn_pack(a, True),
    "tce_ffn_fused_f32": _ffn_fused,
    "tce_ffn_fused_split_f32": _ffn_fused_split,
    "tce_xattn_prepare_f32": lambda a: (
        [dense(_p(a[0]), a[9] * a[7] * 256 * F), dense(_p(a[1]), a[9] * a[7] * 256 * F), dense(_p(a[2]), 257 * 256 * F),
         dense(_p(a[3]), 256 * 256 * F)],
        [dense(_p(a[4]), a[9] * 8 * a[8] * 256 * F), dense(_p(a[5]), a[9] * 8 * a[8] * F), dense(_p(a[6]), a[9] * 8 * a[8] * 256 * F)]),
    "tce_xattn_fused_f32": lambda a: _xattn(_st(a[0])),
    "tce_xattn_ffn_fused_f32": lambda a: _xattn_ffn(_st(a[0]), _st(a[1])),
    "tce_ffn_pack_chain_f32": lambda a: _ffn_pack(a, False),
    # k, v, wqT, wo, packed, L, group, batch
    "tce_xattn_pack_f32": lambda a: (
        [dense(_p(a[0]), a[7] * a[5] * 256 * F), dense(_p(a[1]), a[7] * a[5] * 256 * F), dense(_p(a[2]), 257 * 256 * F), dense(_p(a[3]), 256 * 256 * F)],
        [dense(_p(a[4]), a[7] * _lib.lib_raw().tce_ffn_packed_bytes(256, 8 * a[6]))]),
    "tce_rowlin_pack_f32": lambda a: ([strided(_p(a[0]), a[4] * F, (a[3], a[1] * F))],
                                      [dense(_p(a[2]), _lib.lib_raw().tce_rowlin_packed_bytes(a[3], a[4]))]),
    "tce_rowlin_f32": lambda a: _rowlin(_st(a[0])),
    "tce_conv3x3_pack_f32": lambda a: ([dense(_p(a[0]), a[3] * 9 * a[2] * F)],
                                       [dense(_p(a[1]), _lib.lib_raw().tce_conv3x3_packed_bytes(a[2], a[3]))]),
    "tce_conv3x3_f32": lambda a: (
        [strided(_p(a[0]), a[9] * F, (a[6] * a[7] * a[8], a[1] * F)), dense(_p(a[2]), _lib.lib_raw().tce_conv3x3_packed_bytes(a[9], a[10])),
         dense(_p(a[3]), a[10] * F)],
        [strided(_p(a[4]), a[10] * F, (a[6] * a[7] * a[8], a[5] * F))]),
    "tce_fewrow_linear_f32": lambda a: _fewrow(_st(a[0])),
    # x, ldx, xsplits, bias_x, act_x, W, ldw, ws, M, N, K
    "tce_thin_partials_f32": lambda a: (
        [(strided(_p(a[0]), a[10] * F, (a[8], a[1] * F)) if a[2] == 0 else dense(_p(a[0]), a[2] * a[8] * a[10] * F)),
         dense(_p(a[3]), a[10] * F), strided(_p(a[5]), a[10] * F, (a[9], a[6] * F))],
        [dense(_p(a[7]), (a[10] // 256) * a[8] * a[9] * F)]),
    # ws, splits, M, N, bias, act, res, ldres, res_mode, C, ldc, gamma, beta, eps
    "tce_splitk_reduce_f32": lambda a: (
        [dense(_p(a[0]), a[1] * a[2] * a[3] * F), dense(_p(a[4]), a[3] * F), dense(_p(a[11]), a[3] * F), dense(_p(a[12]), a[3] * F)] +
        ([strided(_p(a[6]), a[3] * F, (a[2], a[7] * F))] if a[8] else []),
        [strided(_p(a[9]), a[3] * F, (a[2], a[10] * F))]),
    # planes, splits, bias, out, L, nheads
    "tce_mha_small64_splits_f32": lambda a: ([dense(_p(a[0]), a[1] * a[4] * 3 * a[5] * 64 * F), dense(_p(a[2]), 3 * a[5] * 64 * F)],
                                             [dense(_p(a[3]), a[4] * a[5] * 64 * F)]),
    # planes, splits, bias, out, nseq, L, nheads
    "tce_mha_small64_seqs_f32": lambda a: ([dense(_p(a[0]), a[1] * a[4] * a[5] * 3 * a[6] * 64 * F), dense(_p(a[2]), 3 * a[6] * 64 * F)],
                                           [dense(_p(a[3]), a[4] * a[5] * a[6] * 64 * F)]),
    # ids, word, pos, type0, gamma, beta, out, nseq, seq_len, C
    "tce_embed_ln_seqs_f32": lambda a: ([dense(_p(a[0]), a[7] * a[8] * 8), dense(_p(a[4]), a[9] * F), dense(_p(a[5]), a[9] * F)],
                                        [dense(_p(a[6]), a[7] * a[8] * a[9] * F)]),
    "tce_swin_attn_pack_f32": lambda a: ([dense(_p(a[0]), 3 * a[3] * a[3] * F), dense(_p(a[1]), a[3] * a[3] * F)],
                                         [dense(_p(a[2]), _lib.lib_raw().tce_swin_attn_packed_bytes(a[3]))]),
    # x, ldx, packed, qkv_bias, proj_bias, table, g1, be1, eps, out, ldo, T, H, W, C, shift
    "tce_swin_attn_fused_f32": lambda a: (
        [strided(_p(a[0]), a[14] * F, (a[11] * a[12] * a[13], a[1] * F)), dense(_p(a[2]), _lib.lib_raw().tce_swin_attn_packed_bytes(a[14])),
         dense(_p(a[3]), 3 * a[14] * F), dense(_p(a[4]), a[14] * F), dense(_p(a[5]), 169 * (a[14] // 32) * F),
         dense(_p(a[6]), a[14] * F), dense(_p(a[7]), a[14] * F)],
        [strided(_p(a[9]), a[14] * F, (a[11] * a[12] * a[13], a[10] * F))]),
}
# Entry points that launch nothing (queries, process switches, graph helpers, tuning aids): passed through.
NOT_LAUNCHES = {"tce_abi_version", "tce_last_error", "tce_gemm_select_tile", "tce_gemm_select_tile_ex", "tce_set_gemm_mode", "tce_set_gemm_mode_thread",
                "tce_get_gemm_mode", "tce_set_range_flag", "tce_groupnorm_nsplit", "tce_mha_ws_bytes", "tce_ffn_packed_bytes", "tce_ffn_split_ws_floats", "tce_ffn_split_counters",
                "tce_rowlin_packed_bytes", "tce_conv3x3_packed_bytes", "tce_swin_attn_packed_bytes", "tce_thin_linear_splits", "tce_graph_begin", "tce_graph_end", "tce_graph_launch",
                "tce_graph_destroy", "tce_graph_group"} | set(_lib.DEBUG_SIGNATURES)


# ---------------------------------------------------------------------------------------------------------------------
# the recorder
# ---------------------------------------------------------------------------------------------------------------------
class Launch:
    __slots__ = ("idx", "name", "stream", "tick", "clock", "reads", "writes", "site")

    def __repr__(self):
        return f"#{self.idx} {self.name} on stream {self.stream:#x} at {self.site}"


class Recorder:
    """Vector clocks over streams: clock[s] maps stream -> the latest tick of that stream known to have completed before
    whatever `s` runs next."""

    def __init__(self):
        self.launches = []
        self.clock = {}      # stream -> {stream: tick}
        self.events = {}     # id(event) -> clock snapshot
        self.edges = 0
        self.host_syncs = 0

    def _clk(self, s):
        c = self.clock.get(int(s))
        if c is None:  # a stream first seen after a host sync still starts behind that sync
            c = self.clock[int(s)] = dict(getattr(self, "_future", {}))
        return c

    def launch(self, name, stream, reads, writes, site):
        s = int(stream or 0)
        c = self._clk(s)
        c[s] = c.get(s, 0) + 1
        L = Launch()
        L.idx, L.name, L.stream, L.tick, L.clock = len(self.launches), name, s, c[s], dict(c)
        L.reads, L.writes, L.site = union(*reads), union(*writes), site
        self.launches.append(L)

    def record_event(self, ev, stream):
        self.events[id(ev)] = (ev, dict(self._clk(stream)))  # the event is kept alive: ids are never reused while recording

    @staticmethod
    def _merge(c, snap):
        for k, v in snap.items():
            if v > c.get(k, 0):
                c[k] = v

    def wait_event(self, ev, stream):
        ent = self.events.get(id(ev))
        if ent is None:
            return
        self._merge(self._clk(stream), ent[1])
        self.edges += 1

    def host_sync(self, snap=None):
        """The host waited for the device.  snap = the clock known complete (an event's snapshot, a stream's clock); None =
        everything issued so far (device synchronize).  What is complete precedes everything issued later, on any stream."""
        if snap is None:
            snap = {}
            for c in self.clock.values():
                self._merge(snap, c)
        snap = dict(snap)
        self._future = getattr(self, "_future", {})
        self._merge(self._future, snap)
        for s in list(self.clock):
            self._merge(self.clock[s], snap)
        self.host_syncs += 1

    @staticmethod
    def ordered(a, b):
        """a issued before b: does a happen-before b?"""
        return b.clock.get(a.stream, 0) >= a.tick

    def analyse(self, max_report=20):
        L = self.launches
        conflicts, pairs = [], 0
        by_stream = {}
        for x in L:
            by_stream.setdefault(x.stream, []).append(x)
        for j, b in enumerate(L):
            for s, lst in by_stream.items():
                if s == b.stream:
                    continue
                seen = b.clock.get(s, 0)
                for a in lst:  # issue order within a stream = tick order
                    if a.idx >= b.idx:
                        break
                    if a.tick <= seen:
                        continue
                    pairs += 1
                    for kind, x, y in (("write/write", a.writes, b.writes), ("write/read", a.writes, b.reads),
                                       ("read/write", a.reads, b.writes)):
                        ov = overlap(x, y)
                        if ov is not None:
                            conflicts.append((kind, a, b, ov))
                            break
        return Report(len(L), len(by_stream), self.edges, self.host_syncs, pairs, conflicts[:max_report], len(conflicts))


class Report:
    def __init__(self, launches, streams, edges, host_syncs, pairs, conflicts, n_conflicts):
        self.launches, self.streams, self.edges, self.host_syncs = launches, streams, edges, host_syncs
        self.unordered_pairs, self.conflicts, self.n_conflicts = pairs, conflicts, n_conflicts

    @property
    def clean(self):
        return self.n_conflicts == 0

    def __str__(self):
        head = (f"hazard check: {self.launches} launches on {self.streams} streams, {self.edges} cross-stream edges, "
                f"{self.host_syncs} host syncs, {self.unordered_pairs} unordered launch pairs compared: "
                f"{'NO conflicting pair (race free)' if self.clean else str(self.n_conflicts) + ' CONFLICTS'}")
        lines = [head]
        for kind, a, b, (lo, hi) in self.conflicts:
            lines.append(f"  {kind} on [{lo:#x}, {hi:#x}) ({hi - lo} bytes):\n    {a}\n    {b}")
        return "\n".join(lines)


def _site():
    """Innermost frame of the launch program (pipeline / text_encoder / model) on the stack: 'file:line function'."""
    best = "?"
    for fr in traceback.extract_stack(limit=24)[:-3]:
        fn = fr.filename.rsplit("/", 1)[-1]
        if fn in ("pipeline.py", "text_encoder.py", "model.py", "video.py"):
            best = f"{fn}:{fr.lineno} {fr.name}"
    return best


class _LibProxy:
    """Stands where the CDLL stands: launches are recorded (then forwarded), everything else is forwarded."""

    def __init__(self, real, rec, dry=False):
        self._real, self._rec, self._dry = real, rec, dry

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name in NOT_LAUNCHES or not name.startswith("tce_"):
            return fn
        model = MODELS.get(name)
        if model is None:
            raise RuntimeError(f"hazard checker: no access model for {name} (add one to hazard.MODELS)")
        rec, dry = self._rec, self._dry

        def call(*a):
            rd, wr = model(a)
            rec.launch(name, a[-1], rd, wr, _site())
            return 0 if dry else fn(*a)  # dry: record only, launch nothing (negative controls with deliberately broken edges)
        return call


@contextlib.contextmanager
def recording(dry=False):
    """with hazard.recording() as rec: ...issue the launch program...; rec.analyse().  dry=True records without launching."""
    rec = Recorder()
    real = _lib.lib()
    Event, Stream = torch.cuda.Event, torch.cuda.Stream
    ev_record, ev_wait, ev_sync, st_sync, dev_sync = Event.record, Event.wait, Event.synchronize, Stream.synchronize, torch.cuda.synchronize

    def record(self, stream=None):
        stream = torch.cuda.current_stream() if stream is None else stream
        rec.record_event(self, stream.cuda_stream)
        return ev_record(self, stream)

    def wait(self, stream=None):
        stream = torch.cuda.current_stream() if stream is None else stream
        rec.wait_event(self, stream.cuda_stream)
        return ev_wait(self, stream)

    def sync_ev(self):
        r = ev_sync(self)
        ent = rec.events.get(id(self))
        if ent is not None:
            rec.host_sync(ent[1])  # only what the event had seen is known complete
        return r

    def sync_st(self):
        r = st_sync(self)
        rec.host_sync(rec._clk(self.cuda_stream))
        return r

    def sync_dev(*a, **k):
        r = dev_sync(*a, **k)
        rec.host_sync()
        return r

    _lib._LIB = _LibProxy(real, rec, dry)
    Event.record, Event.wait, Event.synchronize, Stream.synchronize, torch.cuda.synchronize = record, wait, sync_ev, sync_st, sync_dev
    try:
        yield rec
    finally:
        _lib._LIB = real
        Event.record, Event.wait, Event.synchronize, Stream.synchronize, torch.cuda.synchronize = ev_record, ev_wait, ev_sync, st_sync, dev_sync
