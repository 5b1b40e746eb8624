"""The per-clip forward as one straight-line launch program (reference: models/tce_rvos.py:194-393).

Data layout: every activation is token-major fp32 [tokens, C] with tokens = (frame, y, x) -- the layout both
the Swin blocks ([B, H*W, C]) and the deformable transformer ([N, S, C]) already want, and the channels-last
form the implicit-GEMM convolutions read.  The encoder sequence of a frame is the concatenation of its four
levels (S rows), so `memory` is [T, S, 256] exactly as the reference returns it.  Position maps are frame
independent for un-padded clips and are shared by all frames through frame-batched launches (stride 0).
"""
import os

import torch

from . import ops
from .ops import ACT_GELU, ACT_RELU, ACT_RELU_AFTER_RES, RES_ADD, RES_MUL, gemm_ex

D = 256
NH = 8
# Rows from which the fused FFN kernel (one workgroup = 128 tokens x the whole hidden extent, ~150 us at hidden 2048)
# beats two GEMM launches: it needs about half the chip's CUs busy.
FFN_FUSED_MIN_ROWS = int(__import__("os").environ.get("TCE_FFN_FUSED_MIN_ROWS", 16000))
# Tokens from which the one-launch Swin attention half-block (csrc/swinattn.hip: one workgroup per two 7x7 windows) is used
SWIN_FUSED_MIN_TOKENS = int(os.environ.get("TCE_SWIN_FUSED_MIN_TOKENS", 2000))


def _proj_res_ln(x, wt, bias, resid, M, norm_w, norm_b):
    """resid <- LayerNorm(resid + x W^T + bias)  (N = K = 256).  Many rows: one token-stationary launch (csrc/chain.hip, ROW
    mode) instead of GEMM + LayerNorm; otherwise the two launches."""
    pk = ops.rowlin_lookup(wt, D, D) if M >= FFN_FUSED_MIN_ROWS else None
    if pk is not None:
        ops.rowlin(x, pk, resid, M, D, D, D, D, bias=bias, res=resid, ldres=D, res_mode=RES_ADD, ln_out=(norm_w, norm_b))
        return
    gemm_ex(x, wt, resid, M, D, D, D, D, D, bias=bias, res=resid, ldres=D, res_mode=RES_ADD)
    ops.layernorm(resid, norm_w, norm_b, 1e-5, out=resid)


def _lin(A, x, M, K, w, b, N, **kw):
    out = A(M, N)
    gemm_ex(x, w, out, M, N, K, K, K, N, bias=b, **kw)
    return out


# Environment switches of the launch program, read ONCE at import: they are constants of the process, so every capture of
# the process is built from the same program (none of them needs to be part of a graph key).
#   scheduling A/B switches (results identical): TCE_FEWROW, TCE_TEXT_LATE, TCE_TEXT_EARLY_EDGE, TCE_EARLY_PROJ, TCE_TOKFORK, TCE_ENCFORK,
#       TCE_LAT1_AT; TCE_FEWROW_SITES (bisect aid: which sites take the few-row kernel)
#   DIAGNOSTICS that change what forward() returns -- set by tools/ only, announced with a warning at import:
#       TCE_ABLATE (tools/ablate_times.py): the named stages are SKIPPED, results are garbage (the time that disappears is
#           the stage's share of the critical path); model.forward() tags its output dict with out["ablated"];
#       TCE_TAPS=1 (tools/graph_vs_eager.py): copies of intermediates ride out in out["taps"].
ABLATE = set(filter(None, os.environ.get("TCE_ABLATE", "").split(",")))
TAPS = os.environ.get("TCE_TAPS") == "1"
FEWROW_OK = os.environ.get("TCE_FEWROW", "1") != "0"
FEWROW_SITES = os.environ["TCE_FEWROW_SITES"].split(",") if "TCE_FEWROW_SITES" in os.environ else None
TEXT_LATE = os.environ.get("TCE_TEXT_LATE", "1") != "0"
TEXT_EARLY_EDGE = os.environ.get("TCE_TEXT_EARLY_EDGE", "0") != "0"
EARLY_PROJ = os.environ.get("TCE_EARLY_PROJ", "1") != "0"
TOKFORK = os.environ.get("TCE_TOKFORK", "1") != "0"
ENCFORK = os.environ.get("TCE_ENCFORK", "1") != "0"
DEFER_OUT_NORM = os.environ.get("TCE_DEFER_OUT_NORM", "1") != "0"  # A/B: 0 = every Swin stage's output norm on the backbone's own stream
LAT1_AT = os.environ.get("TCE_LAT1_AT")
SWIN3_FC2_SPLITK = int(os.environ.get("TCE_SWIN3_FC2_SPLITK", 2))
# norm1 / norm2 of the frame-token path in the prologue of the few-row launch that follows them (8 launches fewer per clip): measured
# NOT faster -- 6.055 / 6.047 ms with it against 6.030 / 6.043 without (A/B twice in one call, gpurun_out/r6k): every workgroup of the
# projection recomputes the rows' statistics (two wave reductions per row), which costs what the LayerNorm launch did.  Off.
FTF_LN_FUSE = os.environ.get("TCE_FTF_LN_FUSE", "0") != "0"
if ABLATE or TAPS:
    import warnings
    warnings.warn(f"tce_rvos_amd: DIAGNOSTIC launch program (TCE_ABLATE={sorted(ABLATE)}, TCE_TAPS={int(TAPS)}): "
                  + ("stages are skipped, every result of this process is GARBAGE" if ABLATE else "intermediates are copied out"),
                  RuntimeWarning, stacklevel=2)


_TAP_FN = None  # TCE_TAPS diagnostic: the current clip's tap(name, tensor), set by _run_clip

# Diagnostic: when set to a list, run_clip appends (stage name, event recorded on the main stream at the END of the
# stage) -- tools/stage_times.py (eager mode only).
STAGE_EVENTS = None


def _stage(name):
    if STAGE_EVENTS is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        STAGE_EVENTS.append((name, e))


class _Fork:
    """Runs a block of launches on a side stream (inside hipGraph capture this becomes a parallel graph branch);
    without a side stream it degenerates to in-order execution on the current stream."""

    def __init__(self, side_stream, after=None):
        """after: an event already recorded on the forking stream -- the branch then depends on the work up to THAT point
        only, not on everything issued before the fork (its nodes are still captured / submitted here, in program order)."""
        self.side = side_stream
        self.after = after
        self.ctx = None

    def __enter__(self):
        if self.side is not None:
            if self.after is not None:
                self.side.wait_event(self.after)
            else:
                self.side.wait_stream(torch.cuda.current_stream())
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False

    def join(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)


def run_clip(model, frames, text, img_h, img_w, ar, side_arena=None, side_stream=None, clone_outputs=True, fork2=None,
             fork3=None, valid=None, groups=1, shared=False, select=None):
    """The clip's launch program with the model's own packed-weight routes active (ops.Routes).  valid = (rows, columns) of
    the frames that are not padding (None: un-padded clip).  select = targets[0]['valid_indices'] (tce_rvos.py:233-243)."""
    with ops.routes(model._routes):
        return _run_clip(model, frames, text, img_h, img_w, ar, side_arena, side_stream, clone_outputs, fork2, fork3, valid, groups,
                         shared, select)


def _run_clip(model, frames, text, img_h, img_w, ar, side_arena=None, side_stream=None, clone_outputs=True, fork2=None,
              fork3=None, valid=None, groups=1, shared=False, select=None):
    """text: (last_hidden_state [L,768], pooler_output [768]) or a callable(alloc) returning them (the RoBERTa
    forward, run as a parallel branch beside the backbone when a side stream is given).
    side_arena / side_stream: the decoder (~100 latency-bound launches on 25 rows) runs as a parallel branch
    beside the pixel decoder's large kernels; it needs its own arena because both branches allocate.
    clone_outputs=False returns views into the arenas (valid until the arenas are used again): the graph path owns its
    arenas and copies the outputs out once per replay, so a second copy inside the graph would be wasted.
    fork2 = (arena, stream): a third concurrent branch for the pixel decoder's stride-4 lateral path (see there).
    fork3 = ((arena, stream), (arena, stream)): two more for its stride-32 / stride-16 lateral paths (_pixel_decoder)."""
    cfg, w = model.cfg, model._packed
    dev = frames.device
    # Clip groups (groups = G > 1): `frames` holds G independent clips of Tc frames each, back to back.  Every per-frame /
    # per-token stage simply sees T = G * Tc frames (more rows per launch: the latency-bound stages -- Swin stages 3-4, the text
    # branch, the token and decoder paths -- are shared by the G clips); the stages that look ACROSS a clip's frames or at its
    # caption are block-diagonal per clip: the frame tokens' self-attention, the IQT self-attention, the VisionLanguageBlocks'
    # self-attention, every text cross-attention (one folded weight stream per clip), the decoder's start from the clip's
    # sentence feature.  Each clip's result is the B = 1 forward's (the reference MIXES the clips of a batch).
    # shared = True: the G clips are ONE clip with G captions (the expressions of a video): `frames` holds its Tc frames once, the
    # backbone runs once and its maps are repeated G times; everything after it is the group program above.
    G = int(groups)
    Tb, _, H0, W0 = frames.shape  # frames the backbone sees
    T = Tb * G if shared else Tb
    if T % G:
        raise ValueError("clip group: frames must hold `groups` clips of equal length")
    # select (the A2D / JHMDB single-frame path, tce_rvos.py:233-243): the backbone sees the clip's Tb frames (Video-Swin's
    # windows span them), every stage after it only frame `select` -- the reference index_selects the features, their masks and
    # position maps (frame independent here: same padding in every frame) and continues with t = 1
    if select is not None:
        if G != 1 or not 0 <= int(select) < Tb:
            raise ValueError("valid_indices: one clip per forward, index inside the clip")
        select = int(select)
        T = 1
    Tc = T // G
    ar.reset()
    if side_arena is not None:
        side_arena.reset()
    A = ar.alloc
    sc = model._shape_consts(T, H0, W0, dev, valid)
    lvl_valid = sc["lvl_valid"]  # per-level (rows, columns) that are not padding, or None
    sizes, lvl_sizes, S, starts = sc["sizes"], sc["lvl_sizes"], sc["S"], sc["starts"]
    Q = cfg.num_queries
    ff = cfg.dim_feedforward
    # Few-row kernel (csrc/fewrow.hip): the per-token / per-query projections (a few dozen rows, K = 256) as exact-fp32 launches
    # that serve several projections of the same rows at once (TCE_FEWROW=0: tiled GEMMs).  The LayerNorms stay launches of
    # their own: finishing rows in "the last workgroup" needs an agent-scope release / acquire, on this 8-XCD part an L2
    # write-back + invalidate -- 19 us, more than the launch it would save (profiles/r03_fewrow.txt).
    FR = ops.fewrow_linear

    def few(rows, site=None):  # TCE_FEWROW_SITES: comma list of msda,ftf2,ftf3,dec,text (default: all)
        if FEWROW_SITES is not None and site not in FEWROW_SITES:
            return False
        return FEWROW_OK and rows <= ops.FEWROW_MAX_ROWS

    # ------------------------------------------------------------------ text stage (tce_rvos.py:406-424, FeatureResizer
    # :616-635) and everything that depends on the text alone: the projected keys / values of the five text
    # cross-attention sites and their folded weight streams (csrc/chain.hip, tce_xattn_fused_f32).  With a side stream it
    # is a parallel graph branch beside the backbone; its buffers live in the side arena until the end of the clip.
    tA = (side_arena if side_arena is not None else ar).alloc
    # The text branch needs nothing the backbone produces.  TCE_TEXT_EARLY_EDGE=1 gives it a graph edge back to the START of
    # the clip (its nodes still issued after Swin stage 0's), so that it runs beside stage 0 instead of after it -- measured
    # SLOWER, 8.09 vs 6.90 ms per clip (A/B twice in one gpurun call, profiles/r04_text_early_edge.txt): its ~150 short launches
    # interleave with stage 0's long ones on the same hardware queues and stretch the critical path.  Off by default.
    clip_start = None
    if side_stream is not None and TEXT_EARLY_EDGE:
        clip_start = torch.cuda.Event()
        clip_start.record()
    text_fork = _Fork(side_stream, after=clip_start)
    L = fk = fv = fpk = sent = None
    vl_sites = {}
    text_in = text

    def text_stage():
        nonlocal L, fk, fv, fpk, sent
        with text_fork, model.arith("text"):
            if "text" in ABLATE and callable(text_in):  # diagnostic: the RoBERTa layers skipped (garbage features)
                text_hidden, text_pooled = tA(32, cfg.text_hidden), tA(cfg.text_hidden)
            else:
                text_hidden, text_pooled = text_in(tA) if callable(text_in) else text_in
            GL = text_hidden.shape[0]  # G captions of L tokens, caption-major
            if GL % G:
                raise ValueError("clip group: the text features must hold `groups` captions of equal length")
            L = GL // G
            TH = cfg.text_hidden
            if text_pooled.data_ptr() == text_hidden.data_ptr() + GL * TH * 4 and text_hidden.is_contiguous():
                # the text plan hands hidden states and pooled vectors over as one [G*L + G, 768] tensor: one resizer launch pair
                tmp = _lin(tA, text_hidden, GL + G, TH, w["resizer.fc.weight"], w["resizer.fc.bias"], D)
                both = ops.layernorm(tmp, w["resizer.layer_norm.weight"], w["resizer.layer_norm.bias"], 1e-12, out=tA(GL + G, D))
                text, sent = both[:GL], both[GL:]
            else:
                tmp = _lin(tA, text_hidden, GL, TH, w["resizer.fc.weight"], w["resizer.fc.bias"], D)
                text = ops.layernorm(tmp, w["resizer.layer_norm.weight"], w["resizer.layer_norm.bias"], 1e-12, out=tA(GL, D))
                tmp = _lin(tA, text_pooled, G, TH, w["resizer.fc.weight"], w["resizer.fc.bias"], D)
                sent = ops.layernorm(tmp, w["resizer.layer_norm.weight"], w["resizer.layer_norm.bias"], 1e-12, out=tA(G, D))
            text_pos = model._text_pos(L, dev)
            xattn_ok = L <= 32 and ops.get_gemm_mode() != "f32"

            def text_site(pre, rows, group):
                """(k, v, folded stream or None) of the cross-attention module `pre` whose largest launch has `rows` rows.
                The folded stream is packed in the arithmetic of the site group that consumes it."""
                k = tA(GL, D)  # [G][L, D]: every caption's keys (the position map is shared by the captions)
                if few(GL, "text"):  # key (text + position) and value projections of the site in one launch
                    v = tA(GL, D)
                    FR(text, GL, D, [(w[pre + "k.w"], w[pre + "k.b"], k, D, D, True, 0), (w[pre + "v.w"], w[pre + "v.b"], v, D, D, False, 0)],
                       a2=text_pos, lda2=D, a2_rows=L)
                else:
                    gemm_ex(text, w[pre + "k.w"], k, L, D, D, D, D, D, bias=w[pre + "k.b"], a2=text_pos, lda2=D, batch=G, sA=L * D,
                            sA2=0, sC=L * D)
                    v = _lin(tA, text, GL, D, w[pre + "v.w"], w[pre + "v.b"], D)
                pk = None
                if xattn_ok and rows >= ops.XATTN_MIN_ROWS:
                    with model.arith(group):  # one folded weight stream per caption
                        pk = ops.xattn_pack(k, v, w[pre + "q.wT:x"], w[pre + "out_proj.weight"], L, tA, batch=G)
                return k, v, pk

            fk, fv, fpk = text_site("fusion_module.multihead_attn.", Tc * lvl_sizes[0][0] * lvl_sizes[0][1], "input_proj")
            if cfg.vlblock:
                for stage in (4, 3, 2, 1):
                    h_, w_ = sizes[stage - 1]
                    vl_sites[stage] = text_site(f"pixel_decoder.cross_attn_{stage}.multihead_attn.", Tc * h_ * w_, "pixel.xattn")


    # Capture order = submission order of a replay (the graph's nodes are enqueued in the order they were captured): the ~140
    # few-microsecond launches of the text branch go in AFTER the first Swin stage's, so the backbone (the critical path)
    # does not queue behind them (TCE_TEXT_LATE=0: text first).
    text_late = TEXT_LATE and not cfg.is_resnet
    if not text_late:
        text_stage()

    _stage("start")
    # ------------------------------------------------------------------ backbone
    def ln_(x, pre):
        return ops.layernorm(x, w[pre + ".weight"], w[pre + ".bias"], 1e-5, out=x)

    fused_ok = ops.get_gemm_mode() != "f32"  # the fused kernels ARE split-fp16 arithmetic; exact-fp32 mode = GEMM path

    def ffn(x, M, pre, l1="linear1", l2="linear2", ar=ar, norm=None, group="encoder.ffn"):
        if "ffn:" + group in ABLATE:
            return
        with model.arith(group):
            _ffn(x, M, pre, l1, l2, ar, norm)

    def _ffn(x, M, pre, l1, l2, ar, norm):
        """x <- LN_norm?(x + W2 relu(W1 x)) (in place).  Large M: one fused launch, the [M, 2048] hidden stays on chip
        (csrc/chain.hip); small M (a workgroup walks the whole hidden extent alone): two GEMMs + LayerNorm."""
        pk = w.get(pre + "ffn:pk:" + ops.get_gemm_mode()) if fused_ok else None
        if pk is not None and M >= FFN_FUSED_MIN_ROWS:
            nws, ncnt = ops.ffn_split_need(M, D, ff, ACT_RELU)
            m1 = ar.mark()
            ops.ffn_fused(x, pk, w[pre + l2 + ".bias"], ff, ACT_RELU, M=M,
                          ln_out=(w[norm + ".weight"], w[norm + ".bias"]) if norm else None,
                          split=(ar.alloc(nws), ar.alloc_flags(ncnt)) if nws else None)
            ar.release(m1)
            return
        A = ar.alloc
        m1 = ar.mark()
        hdn = A(M, ff)
        gemm_ex(x, w[pre + l1 + ".weight"], hdn, M, ff, D, D, D, ff, bias=w[pre + l1 + ".bias"], act=ACT_RELU)
        sk = ops.splitk_for(M, D, ff)  # the decoder / frame-token FFNs run on a few dozen rows
        gemm_ex(hdn, w[pre + l2 + ".weight"], x, M, D, ff, ff, ff, D, bias=w[pre + l2 + ".bias"], res=x, ldres=D,
                res_mode=RES_ADD, splitk=sk, ws=A(sk * M * D) if sk > 1 else None,
                ln=(w[norm + ".weight"], w[norm + ".bias"]) if norm else None)  # the norm rides in the split-K reduction
        ar.release(m1)

    def chain_ffn(pre, group, M, norm, A, sMid):
        """The FFN stage of a cross-attention -> FFN chain launch (ops.xattn_fused(ffn=...)), or None where the two run as two
        launches: the chain needs the chain-ordered W1 stream (model._pack: both site groups in one arithmetic) and the row count
        of the fused FFN route.  `mid` (the attention stage's rows, the FFN's residual) is scratch of the caller's arena."""
        if not (ops.XATTN_FFN_CHAIN and fused_ok and M >= FFN_FUSED_MIN_ROWS) or "ffn:" + group in ABLATE:
            return None
        pkc = w.get(pre + "ffn:pkc:" + ops.get_gemm_mode())
        if pkc is None or model.mode_of(group) != ops.get_gemm_mode():
            return None
        return (pkc, w[pre + "linear2.bias"], ff, (w[norm + ".weight"], w[norm + ".bias"]), A(M, D), sMid)

    ffn.chain = chain_ffn  # travels with `ffn` into the pixel decoder's branches

    # input_proj + early fusion of a level (:258-307) needs that level's backbone map and the text only: with the extra
    # branches of fork3 the two large levels start as soon as their Swin stage is done, beside the later stages (which
    # work on few tokens); the two small ones follow the backbone, each on its own stream.
    chs = cfg.num_channels
    early = (fork3 is not None and not cfg.is_resnet and fork3[0][1] is not None and fork3[1][1] is not None and
             side_stream is not None and EARLY_PROJ)
    src = A(T * S, D) if early else None  # [T, S, 256]: the encoder sequence
    lvl_forks = []

    def input_level(l, feat, A):
        with model.arith("input_proj"):
            _input_level(l, feat, A)

    def _input_level(l, feat, A):
        h, ww = lvl_sizes[l]
        hw = h * ww
        if l < 3:
            s = _lin(A, feat, T * hw, chs[1 + l], w[f"input_proj.{l}.0.weight"], w[f"input_proj.{l}.0.bias"], D)
        else:
            h5, w5 = sizes[3]
            # 300 output rows against K = 9*C5 (6912): 20 workgroups would walk the whole K extent one after the other
            k5 = 9 * chs[3]
            sk = next((c for c in (8, 6, 4, 3, 2) if k5 % (c * 32) == 0), 1) if T * h * ww <= 2048 else 1
            s, ho, wo = ops.conv2d_cl(feat, w["input_proj.3.0.weight:cl"], T, h5, w5, chs[3], 3, 3, 2, 1,
                                      bias=w["input_proj.3.0.bias"], alloc=A, splitk=sk,
                                      ws=A(sk * T * h * ww * D) if sk > 1 else None)
            assert (ho, wo) == (h, ww)
        s = ops.groupnorm_cl(s, w[f"input_proj.{l}.1.weight"], w[f"input_proj.{l}.1.bias"], T, hw, D, 32, alloc=A)
        if fpk is not None and Tc * hw >= ops.XATTN_MIN_ROWS:
            # src_l = s * MHA(s, text): straight into the level slice of [T, S, 256]; the Tc frames of a clip share its stream
            ops.xattn_fused(s, fpk, w["fusion_module.multihead_attn.out_proj.bias"], hw, src[starts[l]:], res_mode=RES_MUL,
                            batch=T, sX=hw * D, sRes=hw * D, sOut=S * D, per_batch_weights=G > 1, w_div=Tc)
        else:
            q = _lin(A, s, T * hw, D, w["fusion_module.multihead_attn.q.w"], w["fusion_module.multihead_attn.q.b"], D)
            att = A(T * hw, D)
            ops.mha_core(q, fk, fv, G, NH, Tc * hw, L, D, D, D, Tc * hw * D, L * D, L * D, att, D, Tc * hw * D)
            # src_l = s * out_proj(att), written straight into the level slice of [T, S, 256]
            gemm_ex(att, w["fusion_module.multihead_attn.out_proj.weight"], src[starts[l]:], hw, D, D, D, D, D,
                    bias=w["fusion_module.multihead_attn.out_proj.bias"], res=s, ldres=D, res_mode=RES_MUL, batch=T,
                    sA=hw * D, sC=S * D, sRes=hw * D)

    ar2, stream2 = fork2 if fork2 is not None else (None, None)
    lat1 = None
    if early:
        for arx, _ in fork3:
            arx.reset()

    def pick(feat, i):
        """the selected frame's rows of stage i's map (valid_indices), else the map"""
        if select is None:
            return feat
        hw_i = sizes[i][0] * sizes[i][1]
        return feat[select * hw_i:(select + 1) * hw_i]

    pending_norm0 = []  # stage 0's output norm, applied at the head of the stride-4 lateral branch (start_lat1)

    def on_stage(i, feat, finish=None):
        if i == 0 and text_late:
            text_stage()
        if i == 0 and finish is not None:
            pending_norm0.append(finish)
        feat = pick(feat, i)
        if early:
            if i in (1, 2):
                    arx, stx = fork3[i - 1]
                    fk_ = _Fork(stx)
                    with fk_:
                        if finish is not None:
                            finish()  # the stage's output norm, in this branch
                        text_fork.join()  # this level's stream waits for the text branch (keys / values of the fusion)
                        input_level(i - 1, feat, arx.alloc)
                    lvl_forks.append(fk_)
    if DEFER_OUT_NORM:
        # stage 0's map is read by the stride-4 lateral branch only (started later, on its own stream); stages 1 / 2 by their
        # level's early input projection (forked right here); stage 3 by the main stream itself
        on_stage.defers = lambda i: select is None and ((i == 0 and ar2 is not None and stream2 is not None) or (i in (1, 2) and early))
    if cfg.is_resnet:
        with model.arith("backbone.merge"):  # a ResNet is convolutions only: one site group
            feats = _resnet_backbone(model, frames, ar, sizes, rep=G if shared else 1)
    else:
        feats = _swin_backbone(model, frames, ar, sizes, on_stage, 1 if shared else G, rep=G if shared else 1)
    feats = [pick(f, i) for i, f in enumerate(feats)]

    _stage("backbone")
    text_fork.join()
    _stage("text join")

    # ------------------------------------------------------------------ pixel decoder, stride-4 lateral branch, EARLY
    # adapter_1 + GroupNorm + VisionLanguageBlock at stride 4 read the backbone's C2 map and the text only -- not the
    # encoder memory (segmentation.py:187-196 with the backbone feature as the last lateral input) -- so this millisecond
    # of work is a parallel graph branch, joined when the top-down chain reaches stride 4.
    def start_lat1():
        ar2.reset()
        lat1_fork = _Fork(stream2)
        with lat1_fork:
            while pending_norm0:
                pending_norm0.pop()()  # Swin stage 0's output norm, here where its only consumer runs
            return (lat1_fork, _lateral(model, sc, feats, None, vl_sites, T, L, ffn, ln_, 1, ar2, G))
    # When it starts is a trade: its long kernels (72000-row cross-attention, FFN, GEMMs) slow down whatever runs beside
    # them, and it must be done when the chain reaches stride 4.  Measured at config 2 (ms per clip): right after Swin
    # stage 0 8.09, after the backbone 7.44, after encoder layer 0 / 1 / 2 / 3 (of 4) 7.41 / 7.37 / 7.27 / 7.71 ->
    # one encoder layer before the end ("backbone" / "encN" override for experiments).
    lat1_when = LAT1_AT or (f"enc{cfg.enc_layers - 2}" if cfg.enc_layers >= 2 else "backbone")
    if ar2 is not None and stream2 is not None and lat1_when == "backbone":
        lat1 = start_lat1()
    # ------------------------------------------------------------------ input_proj + early fusion (:258-307)
    # The four levels are independent (they write disjoint slices of src).  Without the early start: level 0 on the main
    # stream, levels 1-3 as one parallel branch on the side stream (idle here: the text branch has joined, the decoder
    # branch has not started).  No buffer is released before the branches have joined.
    if src is None:
        src = A(T * S, D)
    m_levels = ar.mark()
    if early:
        todo = ((2, side_stream), (3, None))
    else:
        todo = ((1, side_stream), (2, side_stream), (3, side_stream), (0, None))
    for l, lvl_stream in todo:
        fk_ = _Fork(lvl_stream)
        lvl_forks.append(fk_)
        with fk_:
            input_level(l, feats[min(1 + l, 3)], A)
    for fk_ in lvl_forks:
        fk_.join()
    ar.release(m_levels)

    _stage("input_proj + fusion")
    # ------------------------------------------------------------------ encoder (:611-627)
    lvl_pos, enc_ref = sc["lvl_pos"], sc["enc_ref"]
    Fk = cfg.f_token
    if Fk > 0:
        token = ops.tile(w["transformer.encoder.memory_bus"], T, out=A(T * Fk, D))
        tpos = w["transformer.encoder.memory_pos"]
        # norm1 / norm2 of the frame-token layer ride in the prologue of the few-row launch that follows them (qk|v of the token
        # self-attention; k|v of the pixel <- token attention): the normalised rows ping-pong between two buffers (a prologue
        # cannot normalise in place: other workgroups still read the rows).  2 launches fewer per layer on the path between layers.
        ftf_fuse_ln = FTF_LN_FUSE and D == 256 and all(few(T * Fk, site) for site in ("msda", "ftf2", "ftf3"))
        token_alt = A(T * Fk, D) if ftf_fuse_ln else None
    # free between the text join and the decoder fork
    tok_stream = side_stream if TOKFORK else None

    def msda(pre, query, q_rows, q_per_frame, q_pos, q_pos_shared, value_src, ref, ref_dim, ref_per_frame, resid,
             ar=ar, norm=None, small_fork=None, group="encoder.msda", defer_norm=False):
        with model.arith(group):
            _msda(pre, query, q_rows, q_per_frame, q_pos, q_pos_shared, value_src, ref, ref_dim, ref_per_frame, resid, ar, norm,
                  small_fork, defer_norm)

    def _msda(pre, query, q_rows, q_per_frame, q_pos, q_pos_shared, value_src, ref, ref_dim, ref_per_frame, resid,
              ar, norm, small_fork, defer_norm=False):
        """resid <- LN_norm?(resid + output_proj(MSDA(query + q_pos, ref, value_proj(value_src)))).  query [T*q_per_frame, D].
        ref: the reference points, or (W, b) of the Linear whose sigmoid gives them (evaluated beside value_proj when
        small_fork is a _Fork: the few-row projections of the frame-token path run as a parallel branch next to the large
        value projection)."""
        A = ar.alloc
        m1 = ar.mark()
        proj = A(q_rows, 384)
        vw = w[pre + "value_proj.weight"]
        # a few queries per frame (frame tokens, decoder queries): "sample, then project" (csrc/msda.hip, msda_fewq_raw_kernel)
        use_raw = ops.MSDA_RAW and q_per_frame <= 64 and T * q_per_frame * NH <= 65536 and vw.is_contiguous()
        if few(q_rows, "msda") and q_pos_shared and norm:
            # a few dozen queries: offsets|weights (+ the reference points' Linear + sigmoid) in one launch
            ref_t = ref
            segs = [(w[pre + "offaw.weight"], w[pre + "offaw.bias"], proj, 384, 384, True, ops.FR_NONE)]
            if isinstance(ref, tuple):
                ref_t = A(q_rows, 2)
                segs.append((ref[0], ref[1], ref_t, 2, 2, False, ops.FR_SIGMOID))
            if use_raw:
                # a few queries per frame: sample the un-projected rows and apply value_proj to the samples -- the [T*S, 256] x
                # [256, 256] projection of the frame (24 us at config 2, on the path between two encoder layers) is not computed
                FR(query, q_rows, D, segs, a2=q_pos, lda2=D, a2_rows=q_per_frame)
                samp = ops.msda_fewq_raw(value_src, vw, w[pre + "value_proj.bias"], proj, ref_t, lvl_sizes, T, S, q_per_frame, 4, 4,
                                         ref_dim, ref_per_frame, out=A(q_rows, D), valid_hw=lvl_valid)
            else:
                fk_s = small_fork if small_fork is not None else _Fork(None)
                with fk_s:
                    FR(query, q_rows, D, segs, a2=q_pos, lda2=D, a2_rows=q_per_frame)
                value = _lin(A, value_src, T * S, D, vw, w[pre + "value_proj.bias"], D)
                fk_s.join()
                samp = ops.msda_fused(value, proj, ref_t, lvl_sizes, T, S, NH, q_per_frame, 4, 4, ref_dim, ref_per_frame,
                                      out=A(q_rows, D), valid_hw=lvl_valid)
            FR(samp, q_rows, D, [(w[pre + "output_proj.weight"], w[pre + "output_proj.bias"], resid, D, D, False, ops.FR_NONE)],
               res=resid, ldres=D)
            if not defer_norm:  # (deferred: the LayerNorm rides in the prologue of the next few-row launch on these rows)
                ln_(resid, norm)
            ar.release(m1)
            return
        if isinstance(ref, tuple):
            ref_lin = ref

            def ref():
                r = _lin(A, query, q_rows, D, ref_lin[0], ref_lin[1], 2)
                return ops.sigmoid(r, out=A(q_rows, 2))

        def offaw():
            if q_pos_shared:  # position map shared by all frames: frame-batched launch, stride 0 on the addend
                gemm_ex(query, w[pre + "offaw.weight"], proj, q_per_frame, 384, D, D, D, 384, bias=w[pre + "offaw.bias"],
                        a2=q_pos, lda2=D, batch=T, sA=q_per_frame * D, sA2=0, sC=q_per_frame * 384)
            else:
                gemm_ex(query, w[pre + "offaw.weight"], proj, q_rows, 384, D, D, D, 384, bias=w[pre + "offaw.bias"],
                        a2=q_pos, lda2=D)
        if use_raw:
            ref = ref() if callable(ref) else ref
            offaw()
            samp = ops.msda_fewq_raw(value_src, vw, w[pre + "value_proj.bias"], proj, ref, lvl_sizes, T, S, q_per_frame, 4, 4,
                                     ref_dim, ref_per_frame, out=A(q_rows, D), valid_hw=lvl_valid)
        else:
            if small_fork is not None:
                with small_fork:
                    ref = ref() if callable(ref) else ref
                    offaw()
            elif callable(ref):
                ref = ref()
            value = _lin(A, value_src, T * S, D, vw, w[pre + "value_proj.bias"], D)
            if small_fork is not None:
                small_fork.join()
            else:
                offaw()
            samp = ops.msda_fused(value, proj, ref, lvl_sizes, T, S, NH, q_per_frame, 4, 4, ref_dim, ref_per_frame,
                                  out=A(q_rows, D), valid_hw=lvl_valid)
        if norm:
            _proj_res_ln(samp, w[pre + "output_proj.weight"], w[pre + "output_proj.bias"], resid, q_rows,
                         w[norm + ".weight"], w[norm + ".bias"])
        else:
            gemm_ex(samp, w[pre + "output_proj.weight"], resid, q_rows, D, D, D, D, D, bias=w[pre + "output_proj.bias"],
                    res=resid, ldres=D, res_mode=RES_ADD)
        ar.release(m1)

    taps = {} if TAPS else None  # bisect aid: copies of intermediates ride out with the outputs

    tap_pool = A(cfg.enc_layers * (16 * T * max(Fk, 1) * D + 3 * T * S * D) + cfg.dec_layers * T * Q * D +
                 2 * T * sum(h_ * w_ for h_, w_ in sizes) * D) if taps is not None else None  # persistent (base level)
    tap_off = [0]

    def tap(name, t):
        if taps is not None:
            n = t.numel()
            buf = tap_pool[tap_off[0]:tap_off[0] + n]
            tap_off[0] += n
            taps[name] = ops.tile(t.reshape(-1), 1, out=buf).view(t.shape)

    global _TAP_FN  # module-level functions (_lateral) tap through it; diagnostic runs only
    _TAP_FN = tap if taps is not None else None

    for i in range(cfg.enc_layers):
        lp = f"transformer.encoder.layers.{i}."
        if Fk > 0:
            fp = lp + "ftoken_layers."
            m0 = ar.mark()
            # The token path is ~16 launches on T*F (= 40) rows, each a kernel boundary long and on the critical path
            # between two encoder layers: independent ones run as parallel graph branches (tok_stream).
            # (1) tokens gather from their frame by MSDA (:447-454); reference points / offsets beside value_proj
            token_ref = (w[fp + "reference_points.weight"], w[fp + "reference_points.bias"])
            if "ftf_tok" not in ABLATE:
                msda(fp + "token_frame_atten.", token, T * Fk, Fk, tpos, True, src, token_ref, 2, True, token,
                     norm=fp + "norm1", small_fork=_Fork(tok_stream), defer_norm=ftf_fuse_ln)
            if not ftf_fuse_ln:
                tap(f"L{i}.token1", token)
            # (2) all T*F tokens attend to each other (:463-469)
            with model.arith("encoder.ftf"):
                R = T * Fk
                if "ftf_tok" not in ABLATE:
                    pre = fp + "token_self_atten."
                    qk = A(R, 2 * D)
                    if few(R, "ftf2"):  # q|k (token + position) and v in one launch
                        v = A(R, D)
                        segs = [(w[pre + "qk.w"], w[pre + "qk.b"], qk, 2 * D, 2 * D, True, ops.FR_NONE),
                                (w[pre + "v.w"], w[pre + "v.b"], v, D, D, False, ops.FR_NONE)]
                        if ftf_fuse_ln:  # token <- norm1(token) in this launch's prologue
                            FR(token, R, D, segs, a2=tpos, lda2=D, a2_rows=Fk, ln_in=(w[fp + "norm1.weight"], w[fp + "norm1.bias"]),
                               xn_out=token_alt)
                            token, token_alt = token_alt, token
                            tap(f"L{i}.token1", token)
                        else:
                            FR(token, R, D, segs, a2=tpos, lda2=D, a2_rows=Fk)
                    else:
                        fk_v = _Fork(tok_stream)
                        with fk_v:
                            v = _lin(A, token, R, D, w[pre + "v.w"], w[pre + "v.b"], D)
                        gemm_ex(token, w[pre + "qk.w"], qk, Fk, 2 * D, D, D, D, 2 * D, bias=w[pre + "qk.b"], a2=tpos, lda2=D,
                                batch=T, sA=Fk * D, sA2=0, sC=Fk * 2 * D)
                        fk_v.join()
                    att = A(R, D)
                    tap(f"L{i}.qk2", qk)
                    tap(f"L{i}.v2", v)
                    Rc = Tc * Fk  # a clip's tokens attend to each other (:463-469), never to another clip's
                    ops.mha_core(qk, qk[:, D:], v, G, NH, Rc, Rc, 2 * D, 2 * D, D, Rc * 2 * D, Rc * 2 * D, Rc * D, att, D, Rc * D)
                    tap(f"L{i}.att2", att)
                    if few(R, "ftf2"):
                        FR(att, R, D, [(w[pre + "out_proj.weight"], w[pre + "out_proj.bias"], token, D, D, False, ops.FR_NONE)],
                           res=token, ldres=D)
                    else:
                        gemm_ex(att, w[pre + "out_proj.weight"], token, R, D, D, D, D, D, bias=w[pre + "out_proj.bias"],
                                res=token, ldres=D, res_mode=RES_ADD)
                    if not ftf_fuse_ln:
                        ln_(token, fp + "norm2")
                if not ftf_fuse_ln:
                    tap(f"L{i}.token2", token)
                # (3) every pixel attends to the F tokens of its own frame (:480-484)
                pre = fp + "frame_token_atten."
                k = A(R, D)
                if few(R, "ftf3"):
                    v = A(R, D)
                    segs = [(w[pre + "k.w"], w[pre + "k.b"], k, D, D, True, ops.FR_NONE),
                            (w[pre + "v.w"], w[pre + "v.b"], v, D, D, False, ops.FR_NONE)]
                    if ftf_fuse_ln and "ftf_tok" not in ABLATE:  # token <- norm2(token) in this launch's prologue
                        FR(token, R, D, segs, a2=tpos, lda2=D, a2_rows=Fk, ln_in=(w[fp + "norm2.weight"], w[fp + "norm2.bias"]),
                           xn_out=token_alt)
                        token, token_alt = token_alt, token
                        tap(f"L{i}.token2", token)
                    else:
                        FR(token, R, D, segs, a2=tpos, lda2=D, a2_rows=Fk)
                else:
                    fk_v = _Fork(tok_stream)
                    with fk_v:
                        v = _lin(A, token, R, D, w[pre + "v.w"], w[pre + "v.b"], D)
                    gemm_ex(token, w[pre + "k.w"], k, Fk, D, D, D, D, D, bias=w[pre + "k.b"], a2=tpos, lda2=D, batch=T,
                            sA=Fk * D, sA2=0, sC=Fk * D)
                    fk_v.join()
            tap(f"L{i}.k3", k)
            tap(f"L{i}.v3", v)
            chained = None
            with model.arith("encoder.ftf_x"):
                if Fk == 8 and fused_ok and T * S >= ops.XATTN_MIN_ROWS:
                    # q-proj -> attention over the frame's 8 tokens -> out-proj -> + src -> norm3 in one token-stationary
                    # launch (the keys / values differ per frame: one folded weight stream per frame)
                    pk = ops.xattn_pack(k, v, w[pre + "q.wT:x"], w[pre + "out_proj.weight"], Fk, A, group=8, batch=T)
                    chained = chain_ffn(fp, "encoder.ffn", T * S, fp + "norm4", A, S * D)
                    ops.xattn_fused(src, pk, w[pre + "out_proj.bias"], S, src, a2=lvl_pos,
                                    ln_out=(w[fp + "norm3.weight"], w[fp + "norm3.bias"]), batch=T, sX=S * D, sOut=S * D,
                                    group=8, per_batch_weights=True, ffn=chained)
                else:
                    q = A(T * S, D)
                    gemm_ex(src, w[pre + "q.w"], q, S, D, D, D, D, D, bias=w[pre + "q.b"], a2=lvl_pos, lda2=D, batch=T,
                            sA=S * D, sA2=0, sC=S * D)
                    att = A(T * S, D)
                    ops.mha_core(q, k, v, T, NH, S, Fk, D, D, D, S * D, Fk * D, Fk * D, att, D, S * D)
                    _proj_res_ln(att, w[pre + "out_proj.weight"], w[pre + "out_proj.bias"], src, T * S, w[fp + "norm3.weight"],
                                 w[fp + "norm3.bias"])
            tap(f"L{i}.src3", src if chained is None else chained[4])
            ar.release(m0)
            # (4) FFN over all pixels (:489-491)
            if chained is None:
                ffn(src, T * S, fp, norm=fp + "norm4")
            tap(f"L{i}.src4", src)
        if "enc_msda" not in ABLATE:
            # the offsets|weights projection (src + pos) and the value projection (src) are independent 24100-row GEMMs of 1.1 and
            # 0.74 workgroup rounds: side by side they pack into 1.9 rounds instead of 3 (TCE_ENCFORK=0: one after the other)
            msda(lp + "self_attn.", src, T * S, S, lvl_pos, True, src, enc_ref, 2, False, src, norm=lp + "norm1",
                 small_fork=_Fork(tok_stream) if ENCFORK else None)
        ffn(src, T * S, lp, norm=lp + "norm2")
        tap(f"L{i}.src6", src)
        if ar2 is not None and stream2 is not None and lat1_when == f"enc{i}":
            lat1 = start_lat1()
    memory = src

    _stage("encoder")
    # ------------------------------------------------------------------ decoder (:721-790) + heads (:330-365)
    nl = cfg.dec_layers
    dar = side_arena if side_arena is not None else ar

    def decoder_branch():
        A = dar.alloc
        nl = cfg.dec_layers
        hs = A(nl, T * Q, D)
        inter_ref = A(nl, T * Q, 4)
        logits = A(nl, T * Q, cfg.num_classes)
        vis = A(nl, T * Q, 1) if cfg.vis_loss else None  # visible_embed heads (--vis_loss, tce_rvos.py:336-338)
        logits_done = [False] * nl
        qpos = w["query_embed.weight"]  # [Q, D], shared by all frames
        if few(Q, "dec"):  # Linear + sigmoid in one launch
            r = A(Q, 2)
            FR(qpos, Q, D, [(w["transformer.reference_points.weight"], w["transformer.reference_points.bias"], r, 2, 2, False, ops.FR_SIGMOID)])
        else:
            r = _lin(A, qpos, Q, D, w["transformer.reference_points.weight"], w["transformer.reference_points.bias"], 2)
            r = ops.sigmoid(r, out=A(Q, 2))
        init_ref = ops.tile(r, T, out=A(T * Q, 2))
        tgt = A(T * Q, D)  # every (frame, query) of a clip starts from the clip's sentence feature
        for g_ in range(G):
            ops.tile(sent[g_], Tc * Q, out=tgt[g_ * Tc * Q:(g_ + 1) * Tc * Q])
        ref, ref_dim = init_ref, 2
        for lid in range(0 if "decoder" not in ABLATE else nl, nl):
            lp = f"transformer.decoder.layers.{lid}."
            m0 = dar.mark()
            pre = lp + "self_attn."
            # the layer works in hs[lid]: its first write (self-attention output + residual) goes there with the previous
            # layer's output as the residual, everything after is in place -- no copy of the layer's result into hs
            prev, tgt = tgt, hs[lid]
            qk = A(T * Q, 2 * D)
            if few(T * Q, "dec"):
                v = A(T * Q, D)
                FR(prev, T * Q, D, [(w[pre + "qk.w"], w[pre + "qk.b"], qk, 2 * D, 2 * D, True, ops.FR_NONE),
                                    (w[pre + "v.w"], w[pre + "v.b"], v, D, D, False, ops.FR_NONE)], a2=qpos, lda2=D, a2_rows=Q)
            else:
                gemm_ex(prev, w[pre + "qk.w"], qk, Q, 2 * D, D, D, D, 2 * D, bias=w[pre + "qk.b"], a2=qpos, lda2=D, batch=T,
                        sA=Q * D, sA2=0, sC=Q * 2 * D)
                v = _lin(A, prev, T * Q, D, w[pre + "v.w"], w[pre + "v.b"], D)
            att = A(T * Q, D)
            if cfg.qtrans:
                # IQT (:683): [T, Q, C] fed seq-first: sequence axis = the clip's frames, batch axis = query slots
                for g_ in range(G):
                    r0 = g_ * Tc * Q
                    ops.mha_core(qk[r0:], qk[r0:, D:], v[r0:], Q, NH, Tc, Tc, Q * 2 * D, Q * 2 * D, Q * D, 2 * D, 2 * D, D, att[r0:],
                                 Q * D, D)
            else:
                ops.mha_core(qk, qk[:, D:], v, T, NH, Q, Q, 2 * D, 2 * D, D, Q * 2 * D, Q * 2 * D, Q * D, att, D, Q * D)
            if few(T * Q, "dec"):
                FR(att, T * Q, D, [(w[pre + "out_proj.weight"], w[pre + "out_proj.bias"], tgt, D, D, False, ops.FR_NONE)],
                   res=prev, ldres=D)
            else:
                gemm_ex(att, w[pre + "out_proj.weight"], tgt, T * Q, D, D, D, D, D, bias=w[pre + "out_proj.bias"], res=prev,
                        ldres=D, res_mode=RES_ADD)
            ln_(tgt, lp + "norm2")
            msda(lp + "cross_attn.", tgt, T * Q, Q, qpos, True, memory, ref, ref_dim, True, tgt, ar=dar, norm=lp + "norm1",
                 group="decoder")
            ffn(tgt, T * Q, lp, ar=dar, norm=lp + "norm3", group="decoder")
            if cfg.with_box_refine:
                bp = f"bbox_embed.{lid}.layers."
                t1, t2 = A(T * Q, D), A(T * Q, D)
                if few(T * Q, "dec"):
                    t3 = A(T * Q, 4)
                    # the level's class head reads the same rows: it rides in the box MLP's first launch
                    segs = [(w[bp + "0.weight"], w[bp + "0.bias"], t1, D, D, False, ops.FR_RELU),
                            (w[f"class_embed.{lid}.weight"], w[f"class_embed.{lid}.bias"], logits[lid], cfg.num_classes,
                             cfg.num_classes, False, ops.FR_NONE)]
                    if vis is not None:
                        segs.append((w[f"visible_embed.{lid}.weight"], w[f"visible_embed.{lid}.bias"], vis[lid], 1, 1, False, ops.FR_NONE))
                    FR(tgt, T * Q, D, segs)
                    logits_done[lid] = True
                    FR(t1, T * Q, D, [(w[bp + "1.weight"], w[bp + "1.bias"], t2, D, D, False, ops.FR_RELU)])
                    FR(t2, T * Q, D, [(w[bp + "2.weight"], w[bp + "2.bias"], t3, 4, 4, False, ops.FR_NONE)])
                else:
                    gemm_ex(tgt, w[bp + "0.weight"], t1, T * Q, D, D, D, D, D, bias=w[bp + "0.bias"], act=ACT_RELU)
                    gemm_ex(t1, w[bp + "1.weight"], t2, T * Q, D, D, D, D, D, bias=w[bp + "1.bias"], act=ACT_RELU)
                    t3 = _lin(A, t2, T * Q, D, w[bp + "2.weight"], w[bp + "2.bias"], 4)
                ops.box_refine(t3, ref, out=inter_ref[lid])
                ref, ref_dim = inter_ref[lid], 4
            dar.release(m0)

        # ------------------------------------------------------------------ heads (:330-365)
        # with box refinement bbox_embed[l] IS transformer.decoder.bbox_embed[l] (tce_rvos.py:124), so
        # sigmoid(bbox_embed[l](hs[l]) + inverse_sigmoid(ref_{l-1})) is exactly inter_ref[l].
        for lvl in range(nl):
            if logits_done[lvl]:
                continue
            ci = lvl if cfg.with_box_refine else 0  # without refinement one head is shared by all levels (:127-130)
            gemm_ex(hs[lvl], w[f"class_embed.{ci}.weight"], logits[lvl], T * Q, cfg.num_classes, D, D, D, cfg.num_classes,
                    bias=w[f"class_embed.{ci}.bias"])
            if vis is not None:
                gemm_ex(hs[lvl], w[f"visible_embed.{ci}.weight"], vis[lvl], T * Q, 1, D, D, D, 1, bias=w[f"visible_embed.{ci}.bias"])
        if cfg.with_box_refine:
            return hs, inter_ref, inter_ref, 4, logits, vis
        # no refinement: every layer saw the initial 2-d reference points; boxes come from the shared bbox_embed
        # applied to each level's hs (+ inverse_sigmoid(ref) on xy), tce_rvos.py:330-349
        refs2 = ops.tile(init_ref, nl, out=A(nl, T * Q, 2))
        bp = "bbox_embed.0.layers."
        t1 = A(nl * T * Q, D)
        gemm_ex(hs, w[bp + "0.weight"], t1, nl * T * Q, D, D, D, D, D, bias=w[bp + "0.bias"], act=ACT_RELU)
        t2 = A(nl * T * Q, D)
        gemm_ex(t1, w[bp + "1.weight"], t2, nl * T * Q, D, D, D, D, D, bias=w[bp + "1.bias"], act=ACT_RELU)
        t3 = _lin(A, t2, nl * T * Q, D, w[bp + "2.weight"], w[bp + "2.bias"], 4)
        ops.box_refine(t3, refs2, out=inter_ref)
        return hs, inter_ref, refs2, 2, logits, vis

    npar = cfg.num_gen_params
    dec_fork = _Fork(side_stream if side_arena is not None else None)
    with dec_fork, model.arith("decoder"):
        hs, boxes, mask_refs, ref_ld, logits, vis = decoder_branch()
        tap("dec.hs", hs)
        # controller MLP + parameter packing of the dynamic mask head (:371-373, 536-559): depend on hs only, so they
        # ride in the decoder branch instead of the main chain's tail
        dA = dar.alloc
        c1 = dA(nl * T * Q, D)
        c2 = dA(nl * T * Q, D)
        if few(nl * T * Q, "dec"):
            params = dA(nl * T * Q, npar)
            FR(hs, nl * T * Q, D, [(w["controller.layers.0.weight"], w["controller.layers.0.bias"], c1, D, D, False, ops.FR_RELU)])
            FR(c1, nl * T * Q, D, [(w["controller.layers.1.weight"], w["controller.layers.1.bias"], c2, D, D, False, ops.FR_RELU)])
            FR(c2, nl * T * Q, D, [(w["controller.layers.2.weight"], w["controller.layers.2.bias"], params, npar, npar, False,
                                    ops.FR_NONE)])
        else:
            gemm_ex(hs, w["controller.layers.0.weight"], c1, nl * T * Q, D, D, D, D, D, bias=w["controller.layers.0.bias"],
                    act=ACT_RELU)
            gemm_ex(c1, w["controller.layers.1.weight"], c2, nl * T * Q, D, D, D, D, D, bias=w["controller.layers.1.bias"],
                    act=ACT_RELU)
            params = _lin(dA, c2, nl * T * Q, D, w["controller.layers.2.weight"], w["controller.layers.2.bias"], npar)
        w0f = dA(T, nl * Q * 8, cfg.mask_dim)
        tail = dA(nl, T * Q, 112)
        ops.mask_pack(params, nl, T, Q, cfg.mask_dim, w0f, tail)
        contrast = None
        if cfg.contrastive:  # contrastive_cal (tce_rvos.py:512-521): cos(mean over the S positions of a frame's memory, sentence feature)
            contrast = dA(T)
            ops.contrastive(memory, sent, T, S, D, Tc, contrast, dA(T * 32 * D))

    _stage("decoder fork")
    # ------------------------------------------------------------------ pixel decoder (segmentation.py:175-296)
    while pending_norm0:  # (the stride-4 lateral branch was never started: its input's norm runs here, before the decoder reads it)
        pending_norm0.pop()()
    mask_feats = _pixel_decoder(model, ar, sc, feats, memory, vl_sites, T, L, ffn, ln_, lat1=lat1, par=fork3, G=G)
    dec_fork.join()

    _stage("pixel decoder")
    # ------------------------------------------------------------------ dynamic mask head (:371-380, 426-510)
    h4, w4 = sizes[0]
    m0 = ar.mark()
    gmat = A(T, h4 * w4, nl * Q * 8)  # (not `G`: the closures above read the clip-group count G late-bound)
    with model.arith("mask_head"):
        ops.gemm_batched(mask_feats.view(T, h4 * w4, cfg.mask_dim), w0f, gmat)
    masks = A(nl, T, Q, h4, w4)
    ops.mask_tail(gmat, tail, mask_refs, ref_ld, masks, nl, T, Q, h4, w4, img_h, img_w, 4)

    _stage("mask head (+ decoder join)")
    # ------------------------------------------------------------------ output dict (:360-393); leave the arena
    K = cfg.num_classes
    keep = (lambda t: t.clone()) if clone_outputs else (lambda t: t)
    out = {
        "pred_logits": keep(logits[-1].reshape(1, T, Q, K)),
        "pred_boxes": keep(boxes[-1].reshape(1, T, Q, 4)),
        "pred_masks": keep(masks[-1].reshape(1, T, Q, h4, w4)),
    }
    if vis is not None:
        out["pred_visible"] = keep(vis[-1].reshape(1, T, Q, 1))
    if contrast is not None:
        out["contrastive"] = keep(contrast.reshape(1, T))  # [b, t] (a clip group: the clips' frames back to back, like every output)
    if cfg.aux_loss:
        out["aux_outputs"] = [{"pred_logits": keep(logits[i].reshape(1, T, Q, K)),
                               "pred_boxes": keep(boxes[i].reshape(1, T, Q, 4)),
                               "pred_masks": keep(masks[i].reshape(1, T, Q, h4, w4))} for i in range(nl - 1)]
        if vis is not None:  # _set_aux_loss with outputs_visible (tce_rvos.py:396-404)
            for i in range(nl - 1):
                out["aux_outputs"][i]["pred_visible"] = keep(vis[i].reshape(1, T, Q, 1))
    if not model.training:
        out["reference_points"] = keep(mask_refs[-2].reshape(1, T, Q, ref_ld)[..., :2])
    out["memory"] = keep(memory.reshape(T, S, D))
    if taps:
        out["taps"] = [{k_: keep(v_) for k_, v_ in taps.items()}]
    ar.release(m0)
    return out


def _swin_backbone(model, frames, ar, sizes, on_stage=None, G=1, rep=1):
    """swin_transformer.py:595-617: returns the four normed stage maps, token-major [T*h*w, C_i].
    Video-Swin (video_swin_transformer.py:678-697): same program with the 3-D window kernel, the (1,4,4) patch
    conv applied per frame, stage outputs taken before the merge and WITHOUT an output norm.
    G > 1 (a clip group): the 2-D windows never leave a frame, so the G clips are simply more frames; the 3-D windows span a
    clip's frames, so the 3-D window kernel is launched once per clip on that clip's rows.
    rep > 1 (one clip, `rep` captions): every returned map holds the stage's map `rep` times, back to back."""
    cfg, w = model.cfg, model._packed
    A = ar.alloc
    T = frames.shape[0]
    b = "backbone.0.body."
    C = cfg.embed_dim
    (H, W) = sizes[0]
    x = A(T * H * W, C)
    with model.arith("backbone.merge"):
        ops.patch_embed(frames, w[b + ("patch_embed.proj.weight:2d" if cfg.video else "patch_embed.proj.weight")],
                        w[b + "patch_embed.proj.bias"],
                        w[b + "patch_embed.norm.weight"], w[b + "patch_embed.norm.bias"], out=x)
    feats = []
    for i, depth in enumerate(cfg.depths):
        H, W = sizes[i]
        ntok = T * H * W
        nH = cfg.num_heads[i]
        out_i = A(rep * ntok, C) if (rep > 1 or not cfg.video) else None
        x_next = None
        if i < len(cfg.depths) - 1:
            H2, W2 = sizes[i + 1]
            x_next = A(T * H2 * W2, 2 * C)
        hid = int(C * cfg.mlp_ratio)
        for j in range(depth if f"swin{i}" not in ABLATE else 0):
            p = f"{b}layers.{i}.blocks.{j}."
            m0 = ar.mark()
            pk_attn = None
            if ops.SWIN_FUSED and not cfg.video and ntok >= SWIN_FUSED_MIN_TOKENS:
                pk_attn = w.get(p + "attn:pk:" + model.mode_of("backbone.attn"))
            if pk_attn is not None:  # norm1 -> qkv -> window attention -> proj -> + x in ONE launch, in place on x
                with model.arith("backbone.attn"):
                    ops.swin_attn_fused(x, pk_attn, w[p + "attn.qkv.bias"], w[p + "attn.proj.bias"],
                                        w[p + "attn.relative_position_bias_table"], w[p + "norm1.weight"], w[p + "norm1.bias"],
                                        T, H, W, C, 0 if j % 2 == 0 else cfg.window_size // 2)
            xn = A(ntok, C)
            if pk_attn is None:
                with model.arith("backbone.attn"):  # x += proj(window attention(qkv(norm1 x))) as three launches
                    qkv = A(ntok, 3 * C)
                    pk = ops.rowlin_lookup(w[p + "attn.qkv.weight"], 3 * C, C) \
                        if ((C <= 128 and ntok >= 32768) or (C == 384 and 2048 <= ntok <= ops.ROWLIN384_MAX_ROWS)) else None
                    if pk is not None:  # norm1 -> qkv in one token-stationary launch (LayerNorm prologue)
                        ops.rowlin(x, pk, qkv, ntok, 3 * C, C, C, 3 * C, bias=w[p + "attn.qkv.bias"],
                                   ln_in=(w[p + "norm1.weight"], w[p + "norm1.bias"]))
                    else:
                        ops.layernorm(x, w[p + "norm1.weight"], w[p + "norm1.bias"], out=xn)
                        gemm_ex(xn, w[p + "attn.qkv.weight"], qkv, ntok, 3 * C, C, C, C, 3 * C, bias=w[p + "attn.qkv.bias"])
                    if cfg.video:
                        nc = ntok // G
                        for g_ in range(G):
                            ops.window_attn3d(qkv[g_ * nc:(g_ + 1) * nc], w[p + "attn.qkv.bias"],
                                              w[p + "attn.relative_position_bias_table"], T // G, H, W, C, nH, j % 2 == 1,
                                              out=xn[g_ * nc:(g_ + 1) * nc])
                        att = xn
                    else:
                        att = ops.window_attn(qkv, w[p + "attn.qkv.bias"], w[p + "attn.relative_position_bias_table"], T, H, W,
                                              C, nH, 0 if j % 2 == 0 else cfg.window_size // 2, out=xn)
                    gemm_ex(att, w[p + "attn.proj.weight"], x, ntok, C, C, C, C, C, bias=w[p + "attn.proj.bias"], res=x, ldres=C,
                            res_mode=RES_ADD)
            with model.arith("backbone.mlp"):  # x += fc2(GELU(fc1(norm2 x)))
                pk = w.get(p + "mlp.ffn:pk:" + ops.get_gemm_mode()) if ops.get_gemm_mode() != "f32" else None
                if pk is not None and ntok >= FFN_FUSED_MIN_ROWS:  # norm2 -> fc1 -> GELU -> fc2 -> +x in one launch
                    ops.ffn_fused(x, pk, w[p + "mlp.fc2.bias"], hid, ACT_GELU, M=ntok,
                                  ln_in=(w[p + "norm2.weight"], w[p + "norm2.bias"]))
                else:
                    hdn = A(ntok, hid)
                    pk1 = ops.rowlin_lookup(w[p + "mlp.fc1.weight"], hid, C) if (C == 384 and 2048 <= ntok <= ops.ROWLIN384_MAX_ROWS) else None
                    if pk1 is not None:  # norm2 -> fc1 -> GELU in one token-stationary launch (LayerNorm prologue)
                        ops.rowlin(x, pk1, hdn, ntok, hid, C, C, hid, bias=w[p + "mlp.fc1.bias"], act=ACT_GELU,
                                   ln_in=(w[p + "norm2.weight"], w[p + "norm2.bias"]))
                    else:
                        ops.layernorm(x, w[p + "norm2.weight"], w[p + "norm2.bias"], out=xn)
                        gemm_ex(xn, w[p + "mlp.fc1.weight"], hdn, ntok, hid, C, C, C, hid, bias=w[p + "mlp.fc1.bias"],
                                act=ACT_GELU)
                    # last stage: ~1200 rows against K = 3072 -- 228 workgroups walking 96 K slices each; split-K: 61 -> 41 us
                    sk = next((c for c in (3, 4, 2) if hid % (c * 32) == 0), 1) \
                        if (ntok <= 2048 and hid >= 2048 and ops.SPLITK_ENABLED) else 1
                    if SWIN3_FC2_SPLITK > 1 and C == 384 and 2048 < ntok <= 12000 and hid % (SWIN3_FC2_SPLITK * 32) == 0:
                        # 216 tiles of 128x64 walk 48 K slices each, one workgroup per CU: two K halves = two co-resident workgroups per
                        # CU hiding each other's stalls + one reduce pass: 6.03 -> 6.00 ms per clip (A/B twice in one call, gpurun_out/r6d)
                        sk = SWIN3_FC2_SPLITK
                    gemm_ex(hdn, w[p + "mlp.fc2.weight"], x, ntok, C, hid, hid, hid, C, bias=w[p + "mlp.fc2.bias"], res=x,
                            ldres=C, res_mode=RES_ADD, splitk=sk, ws=A(sk * ntok * C) if sk > 1 else None)
            ar.release(m0)
        if cfg.video:
            feats.append(x if rep == 1 else ops.tile(x, rep, out=out_i))
            finish = None
        else:
            def finish(x=x, i=i, out_i=out_i, ntok=ntok):
                """the stage's OUTPUT norm (swin_transformer.py:608-611): only the consumers of the map need it, the next stage reads
                the un-normed x"""
                ops.layernorm(x, w[f"{b}norm{i}.weight"], w[f"{b}norm{i}.bias"], out=out_i[:ntok])
                if rep > 1:
                    ops.tile(out_i[:ntok], rep - 1, out=out_i[ntok:])
            # a consumer that runs as a parallel branch applies it there (off the backbone's critical path: 8-10 us per stage)
            if on_stage is not None and getattr(on_stage, "defers", None) is not None and on_stage.defers(i):
                pass
            else:
                finish()
                finish = None
            feats.append(out_i)
        if on_stage is not None:
            on_stage(i, feats[-1], finish)  # the stage's map is final (or its norm handed over): work that needs only this map may start
        if x_next is not None:
            m0 = ar.mark()
            p = f"{b}downsamples.{i}." if cfg.video else f"{b}layers.{i}.downsample."
            xm, _, _ = ops.patch_merge_ln(x, w[p + "norm.weight"], w[p + "norm.bias"], T, H, W, C, alloc=A)
            with model.arith("backbone.merge"):
                gemm_ex(xm, w[p + "reduction.weight"], x_next, x_next.shape[0], 2 * C, 4 * C, 4 * C, 4 * C, 2 * C)
            ar.release(m0)
            x = x_next
            C *= 2
    return feats


def _resnet_backbone(model, frames, ar, sizes, rep=1):
    """models/backbone.py:76-85 over torchvision's bottleneck ResNet (row A11): returns layer1..layer4 maps,
    token-major [T*h*w, 256/512/1024/2048].  Each FrozenBatchNorm2d is folded into the convolution before it
    (model._pack_resnet), so a bottleneck is three (block 0: four) GEMM launches: 1x1 + ReLU, 3x3 (carrying the
    stride, "v1.5") + ReLU as implicit GEMM, 1x1 + identity + ReLU in one epilogue."""
    cfg, w = model.cfg, model._packed
    A = ar.alloc
    T = frames.shape[0]
    b = "backbone.0.body."
    full = [A(rep * T * h * ww, c) for (h, ww), c in zip(sizes, cfg.num_channels)]  # rep > 1: each map `rep` times (one clip,
    feats = [f[:T * h * ww] for f, (h, ww) in zip(full, sizes)]                      # `rep` captions: see _swin_backbone)
    m_all = ar.mark()
    stem, H1, W1 = ops.resnet_stem(frames, w[b + "conv1:f"], w[b + "conv1:b"], alloc=A)
    x, H, W = ops.maxpool3x3s2_cl(stem, T, H1, W1, 64, alloc=A)
    assert (H, W) == tuple(sizes[0])
    cin = 64
    for li, blocks in enumerate(cfg.resnet_blocks):
        width = 64 * 2 ** li
        Ho, Wo = sizes[li]
        for j in range(blocks):
            p = f"{b}layer{li + 1}.{j}."
            stride = 2 if (j == 0 and li > 0) else 1
            out = feats[li] if j == blocks - 1 else A(T * Ho * Wo, 4 * width)
            m0 = ar.mark()
            if j == 0:
                idt, _, _ = ops.conv2d_cl(x, w[p + "downsample.0:f"], T, H, W, cin, 1, 1, stride, 0,
                                          bias=w[p + "downsample.0:b"], alloc=A)
            else:
                idt = x
            y1 = A(T * H * W, width)
            gemm_ex(x, w[p + "conv1:f"], y1, T * H * W, width, cin, cin, cin, width, bias=w[p + "conv1:b"], act=ACT_RELU)
            y2, _, _ = ops.conv2d_cl(y1, w[p + "conv2:f"], T, H, W, width, 3, 3, stride, 1, bias=w[p + "conv2:b"],
                                     act=ACT_RELU, alloc=A)
            gemm_ex(y2, w[p + "conv3:f"], out, T * Ho * Wo, 4 * width, width, width, width, 4 * width,
                    bias=w[p + "conv3:b"], act=ACT_RELU_AFTER_RES, res=idt, ldres=4 * width, res_mode=RES_ADD)
            ar.release(m0)
            x, H, W, cin = out, Ho, Wo, 4 * width
    ar.release(m_all)
    if rep > 1:
        for f, part in zip(full, feats):
            ops.tile(part, rep - 1, out=f[part.shape[0]:])
    return full


def _lateral(model, sc, feats, memory, vl_sites, T, L, ffn, ln_, stage, arx, G=1):
    """Lateral branch of one FPN level (segmentation.py:187-196,326-377): 1x1 adapter + GroupNorm(8) + VisionLanguageBlock.
    Returns tgt [T*hw, 256] allocated in `arx` (temporaries released, tgt stays).  Stage 1 reads the backbone map
    (memory may be None), stages 2-4 the encoder memory."""
    cfg, w = model.cfg, model._packed
    sizes, S, starts = sc["sizes"], sc["S"], sc["starts"]
    pd = "pixel_decoder."
    A = arx.alloc
    h, ww = sizes[stage - 1]
    hw = h * ww
    pos = sc["pos"][stage - 1]  # [hw, 256] backbone-level sine map (no level embedding)
    # lateral 1x1 conv (no bias) + GN(8)
    vis = A(T * hw, D)
    with model.arith("pixel.conv"):
        if stage > 1:
            l = stage - 2  # encoder level index: stage 4 <-> level 2 (32x)
            gemm_ex(memory[starts[l]:], w[f"{pd}adapter_{stage}.weight"], vis, hw, D, D, D, D, D, batch=T, sA=S * D,
                    sC=hw * D)
        else:
            c0 = cfg.num_channels[0]
            gemm_ex(feats[0], w[f"{pd}adapter_1.weight"], vis, T * hw, D, c0, c0, c0, D)
    m1 = arx.mark()
    tgt = ops.groupnorm_cl(vis, w[f"{pd}adapter_{stage}.norm.weight"], w[f"{pd}adapter_{stage}.norm.bias"], T, hw, D, 8,
                           out=vis, alloc=A)
    arx.release(m1)
    if not cfg.vlblock:
        return tgt
    bp = f"{pd}cross_attn_{stage}."
    pre = bp + "self_attn."
    red = sc["red"].get(stage)
    normed = False
    with model.arith("pixel.attn"):
        m1 = arx.mark()
        if red is not None:  # spatial-reduction self-attention (segmentation.py:333-361)
            nh_, nw_, pos_low = red
            n_low = T * nh_ * nw_
            x_low = ops.resize_nearest(tgt, T, h, ww, nh_, nw_, D, alloc=A)
            qk = A(n_low, 2 * D)
            gemm_ex(x_low, w[pre + "qk.w"], qk, nh_ * nw_, 2 * D, D, D, D, 2 * D, bias=w[pre + "qk.b"], a2=pos_low,
                    lda2=D, batch=T, sA=nh_ * nw_ * D, sA2=0, sC=nh_ * nw_ * 2 * D)
            v = _lin(A, x_low, n_low, D, w[pre + "v.w"], w[pre + "v.b"], D)
            att = A(n_low, D)
            nc = n_low // G  # a clip's reduced tokens attend to each other only
            ops.mha_core(qk, qk[:, D:], v, G, NH, nc, nc, 2 * D, 2 * D, D, nc * 2 * D, nc * 2 * D, nc * D, att, D, nc * D, alloc=A,
                         kmask=sc["kmask"].get(stage))  # padded clips: padded positions are no keys (segmentation.py:345-356)
            o_low = _lin(A, att, n_low, D, w[pre + "out_proj.weight"], w[pre + "out_proj.bias"], D)
            ops.resize_bilinear(o_low, T, nh_, nw_, h, ww, D, add=tgt, out=tgt, ln=(w[bp + "norm1.weight"], w[bp + "norm1.bias"]))
            normed = True  # norm1 rode in the resize + add pass
        else:
            n = T * hw
            qk = A(n, 2 * D)
            gemm_ex(tgt, w[pre + "qk.w"], qk, hw, 2 * D, D, D, D, 2 * D, bias=w[pre + "qk.b"], a2=pos, lda2=D,
                    batch=T, sA=hw * D, sA2=0, sC=hw * 2 * D)
            v = _lin(A, tgt, n, D, w[pre + "v.w"], w[pre + "v.b"], D)
            att = A(n, D)
            nc = n // G
            ops.mha_core(qk, qk[:, D:], v, G, NH, nc, nc, 2 * D, 2 * D, D, nc * 2 * D, nc * 2 * D, nc * D, att, D, nc * D, alloc=A,
                         kmask=sc["kmask"].get(stage))
            gemm_ex(att, w[pre + "out_proj.weight"], tgt, n, D, D, D, D, D, bias=w[pre + "out_proj.bias"], res=tgt,
                    ldres=D, res_mode=RES_ADD)
        arx.release(m1)
    if not normed:
        ln_(tgt, bp + "norm1")
    # text cross-attention (:366-371)
    pre = bp + "multihead_attn."
    with model.arith("pixel.xattn"):
        m1 = arx.mark()
        tk, tv, pk = vl_sites[stage]
        Mc = (T // G) * hw  # rows of one clip
        chained = None
        if pk is not None:
            # q-proj -> attention over the text keys -> out-proj -> + tgt -> norm2 in one token-stationary launch (one batch entry
            # and one folded weight stream per clip); with the FFN + norm3 behind it in the same launch where chain_ffn allows
            chained = ffn.chain(bp, "pixel.ffn", T * hw, bp + "norm3", A, Mc * D)
            ops.xattn_fused(tgt, pk, w[pre + "out_proj.bias"], Mc, tgt, a2=pos, a2_rows=hw,
                            ln_out=(w[bp + "norm2.weight"], w[bp + "norm2.bias"]), batch=G, sX=Mc * D, sOut=Mc * D,
                            per_batch_weights=G > 1, ffn=chained)
        else:
            q = A(T * hw, D)
            gemm_ex(tgt, w[pre + "q.w"], q, hw, D, D, D, D, D, bias=w[pre + "q.b"], a2=pos, lda2=D, batch=T, sA=hw * D,
                    sA2=0, sC=hw * D)
            att = A(T * hw, D)
            ops.mha_core(q, tk, tv, G, NH, Mc, L, D, D, D, Mc * D, L * D, L * D, att, D, Mc * D)
            _proj_res_ln(att, w[pre + "out_proj.weight"], w[pre + "out_proj.bias"], tgt, T * hw, w[bp + "norm2.weight"],
                         w[bp + "norm2.bias"])
        arx.release(m1)
    if chained is None:
        ffn(tgt, T * hw, bp, norm=bp + "norm3", ar=arx, group="pixel.ffn")
    if _TAP_FN is not None:
        _TAP_FN(f"vl{stage}", tgt)  # the VisionLanguageBlock's output (segmentation.py:326-377)
    return tgt



def _pixel_decoder(model, ar, sc, feats, memory, vl_sites, T, L, ffn, ln_, lat1=None, par=None, G=1):
    """CrossModalFPNDecoder.forward: top-down FPN with a VisionLanguageBlock at every level.

    The lateral branch of a level (_lateral) depends only on its input map and the text; only the top-down merge + 3x3
    convolution chain is sequential (stage 4 -> 1).  lat1 = (fork, tgt) is the stride-4 lateral branch started early by
    run_clip as a parallel graph branch; it is joined when the chain reaches stride 4.  par = ((arena, stream),
    (arena, stream)): the stride-32 lateral + the chain down to stride 16 and the stride-16 lateral run as two more
    parallel branches beside the stride-8 lateral (the largest of the three) on the main stream."""
    cfg, w = model.cfg, model._packed
    A = ar.alloc
    sizes = sc["sizes"]
    pd = "pixel_decoder."

    pending = {}  # id of a y buffer that holds the RAW convolution output of its stage -> that stage (its GroupNorm is still due)

    def merge_conv(tgt, stage, y, y_hw, arx, y_new):
        """top-down merge (nearest up-sampling to the exact finer size) + 3x3 conv + GN(8) + ReLU -> y_new.  The GroupNorm + ReLU
        of a stage whose output only feeds the next merge (stages 4..2) is not applied here: y_new then holds the raw convolution
        output and the NEXT merge applies it while it up-samples and adds (ops.groupnorm_up_add: one apply pass instead of an
        apply pass and a merge pass).  Stage 1's output feeds the mask_features convolution: normalised here."""
        h, ww = sizes[stage - 1]
        if y is not None:
            prev = pending.pop(id(y), None)
            if prev is not None:
                ops.groupnorm_up_add(y, w[f"{pd}layer_{prev}.norm.weight"], w[f"{pd}layer_{prev}.norm.bias"], T, y_hw[0], y_hw[1], h, ww,
                                     D, 8, add=tgt, out=tgt, relu=True, alloc=arx.alloc)
            else:
                ops.resize_nearest(y, T, y_hw[0], y_hw[1], h, ww, D, add=tgt, out=tgt)
        if ops.GN_UP_FUSE and stage > 1:
            with model.arith("pixel.conv"):
                ops.conv2d_cl(tgt, w[f"{pd}layer_{stage}.weight:cl"], T, h, ww, D, 3, 3, 1, 1, out=y_new, alloc=arx.alloc)
            pending[id(y_new)] = stage
            return
        with model.arith("pixel.conv"):
            conv, _, _ = ops.conv2d_cl(tgt, w[f"{pd}layer_{stage}.weight:cl"], T, h, ww, D, 3, 3, 1, 1, alloc=arx.alloc)
        ops.groupnorm_cl(conv, w[f"{pd}layer_{stage}.norm.weight"], w[f"{pd}layer_{stage}.norm.bias"], T, h * ww, D, 8,
                         relu=True, out=y_new, alloc=arx.alloc)

    y = None
    y_hw = None
    stages = (4, 3, 2, 1)
    if par is not None and par[0][1] is not None and par[1][1] is not None:
        (ar3, st3), (ar4, st4) = par
        ar3.reset()
        ar4.reset()
        fk_l3 = _Fork(st4)
        with fk_l3:
            tgt3 = _lateral(model, sc, feats, memory, vl_sites, T, L, ffn, ln_, 3, ar4, G)
        fk_c = _Fork(st3)
        with fk_c:
            (h4, w4), (h3, w3) = sizes[3], sizes[2]
            y4, y3 = ar3.alloc(T * h4 * w4, D), ar3.alloc(T * h3 * w3, D)
            tgt4 = _lateral(model, sc, feats, memory, vl_sites, T, L, ffn, ln_, 4, ar3, G)
            merge_conv(tgt4, 4, None, None, ar3, y4)
            fk_l3.join()  # this branch waits for the stride-16 lateral
            merge_conv(tgt3, 3, y4, (h4, w4), ar3, y3)
        h, ww = sizes[1]
        y2 = A(T * h * ww, D)
        m0 = ar.mark()
        tgt2 = _lateral(model, sc, feats, memory, vl_sites, T, L, ffn, ln_, 2, ar, G)
        fk_c.join()
        merge_conv(tgt2, 2, y3, (h3, w3), ar, y2)
        ar.release(m0)
        y, y_hw = y2, (h, ww)
        stages = (1,)
    for stage in stages:
        h, ww = sizes[stage - 1]
        hw = h * ww
        y_new = A(T * hw, D)
        m0 = ar.mark()
        if stage == 1 and lat1 is not None:
            lat1[0].join()
            tgt = lat1[1]
        else:
            tgt = _lateral(model, sc, feats, memory, vl_sites, T, L, ffn, ln_, stage, ar, G)
        merge_conv(tgt, stage, y, y_hw, ar, y_new)
        ar.release(m0)
        y, y_hw = y_new, (h, ww)
    h, ww = sizes[0]
    with model.arith("pixel.conv"):
        out_final, _, _ = ops.conv2d_cl(y, w[pd + "mask_features.weight:cl"], T, h, ww, D, 3, 3, 1, 1,
                                        bias=w[pd + "mask_features.bias"], alloc=A)
    return out_final
