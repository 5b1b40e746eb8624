"""ctypes binding of libtce_rvos.so (the C ABI declared in include/tce_rvos.h).

The product path has NO fallback: if the shared object is missing or a symbol is absent, importing this
module's `lib()` raises.  Build it with `python -m tce_rvos_amd.build` / `__graft_entry__.build()`.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TCE_LIB") or os.path.join(HERE, "lib", "libtce_rvos.so")  # TCE_LIB: A/B of two builds (tools)

c_f = C.c_void_p  # device pointers travel as integers
i32, i64, f32 = C.c_int32, C.c_int64, C.c_float


class GemmArgs(C.Structure):
    _fields_ = [("A", c_f), ("A2", c_f), ("W", c_f), ("bias", c_f), ("res", c_f), ("C", c_f),
                ("M", i32), ("N", i32), ("K", i32),
                ("lda", i32), ("lda2", i32), ("ldw", i32), ("ldc", i32), ("ldres", i32),
                ("act", i32), ("res_mode", i32), ("batch", i32),
                ("sA", i64), ("sA2", i64), ("sW", i64), ("sBias", i64), ("sC", i64), ("sRes", i64),
                ("conv", i32), ("T", i32), ("H", i32), ("Wd", i32), ("Cin", i32), ("Ho", i32), ("Wo", i32),
                ("kh", i32), ("kw", i32), ("stride", i32), ("pad", i32)]


class RowLinArgs(C.Structure):
    _fields_ = [("x", c_f), ("a2", c_f), ("packed", c_f), ("bias", c_f), ("res", c_f), ("out", c_f),
                ("g_in", c_f), ("be_in", c_f), ("g_out", c_f), ("be_out", c_f),
                ("ldx", i64), ("lda2", i64), ("ldres", i64), ("ldo", i64),
                ("sX", i64), ("sA2", i64), ("sRes", i64), ("sOut", i64),
                ("M", i32), ("N", i32), ("K", i32), ("batch", i32), ("a2_rows", i32), ("act", i32), ("res_mode", i32),
                ("eps_in", f32), ("eps_out", f32)]


class XattnArgs(C.Structure):
    _fields_ = [("x", c_f), ("a2", c_f), ("packed", c_f), ("bo", c_f), ("res", c_f), ("out", c_f),
                ("g_out", c_f), ("be_out", c_f),
                ("ldx", i64), ("lda2", i64), ("ldres", i64), ("ldo", i64),
                ("sX", i64), ("sRes", i64), ("sOut", i64), ("sW", i64),
                ("M", i32), ("batch", i32), ("a2_rows", i32), ("res_mode", i32), ("group", i32), ("eps_out", f32), ("w_div", i32)]



class XattnFfnArgs(C.Structure):
    _fields_ = [("packed", c_f), ("b2", c_f), ("g_out", c_f), ("be_out", c_f), ("mid", c_f), ("ldmid", i64), ("sMid", i64),
                ("hidden", i32), ("act", i32), ("eps_out", f32)]


class FewRowSeg(C.Structure):
    _fields_ = [("W", c_f), ("bias", c_f), ("out", c_f), ("N", i32), ("ldw", i32), ("ldo", i32), ("use_a2", i32), ("act", i32)]


class FewRowArgs(C.Structure):
    _fields_ = [("x", c_f), ("a2", c_f), ("res", c_f),
                ("ldx", i64), ("lda2", i64), ("ldres", i64),
                ("a2_rows", i32), ("R", i32), ("K", i32), ("nseg", i32), ("seg", FewRowSeg * 3),
                ("g_in", c_f), ("be_in", c_f), ("xn_out", c_f), ("ldxn", i64), ("eps_in", f32)]


class CopySeg(C.Structure):
    _fields_ = [("src", c_f), ("dst", c_f), ("rows", i64), ("row_words", i64), ("src_pitch_words", i64)]


# name -> (restype, argtypes); must list EVERY symbol of include/tce_rvos.h (tests check this)
SIGNATURES = {
    "tce_abi_version": (i32, []),
    "tce_last_error": (C.c_char_p, []),
    "tce_gemm_f32": (i32, [C.POINTER(GemmArgs), c_f]),
    "tce_gemm_splitk_f32": (i32, [C.POINTER(GemmArgs), i32, c_f, c_f]),
    "tce_gemm_splitk_ln_f32": (i32, [C.POINTER(GemmArgs), i32, c_f, c_f, c_f, f32, c_f]),
    "tce_gemm_select_tile": (i32, [i32, i32, i32]),
    "tce_gemm_select_tile_ex": (i32, [i32, i32, i32, i32, i32]),
    "tce_set_gemm_mode": (i32, [i32]),
    "tce_set_gemm_mode_thread": (i32, [i32]),
    "tce_set_range_flag": (i32, [c_f]),
    "tce_get_gemm_mode": (i32, []),
    "tce_layernorm_f32": (i32, [c_f, c_f, c_f, c_f, c_f, i64, i32, f32, c_f]),
    "tce_groupnorm_nsplit": (i32, [i32]),
    "tce_groupnorm_f32": (i32, [c_f, c_f, c_f, c_f, c_f, i32, i32, i32, i32, f32, i32, c_f]),
    "tce_groupnorm_up_add_f32": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, i32, i32, i32, i32, i32, f32, i32, c_f]),
    "tce_resnet_stem_f32": (i32, [c_f, c_f, c_f, c_f, i32, i32, i32, c_f]),
    "tce_maxpool3x3s2_cl_f32": (i32, [c_f, c_f, i32, i32, i32, i32, c_f]),
    "tce_resize_h_u8": (i32, [c_f, c_f, c_f, c_f, i64, i32, i32, i32, c_f]),
    "tce_resize_v_norm_f32": (i32, [c_f, c_f, c_f, c_f, c_f, i32, i32, i32, i32, i32, c_f]),
    "tce_patch_embed_f32": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, i32, i32, f32, c_f]),
    "tce_window_attn_f32": (i32, [c_f, c_f, c_f, c_f, i32, i32, i32, i32, i32, i32, c_f]),
    "tce_window_attn3d_f32": (i32, [c_f, c_f, c_f, c_f, i32, i32, i32, i32, i32, i32, c_f]),
    "tce_patch_merge_ln_f32": (i32, [c_f, c_f, c_f, c_f, i32, i32, i32, i32, f32, c_f]),
    "tce_mha_f32": (i32, [c_f, c_f, c_f, c_f, i32, i32, i32, i32, i32, i32, i32, i32, i64, i64, i64, i64, c_f, f32, c_f]),
    "tce_mha_ws_bytes": (i64, [i32, i32, i32]),
    "tce_mha_ws_f32": (i32, [c_f, c_f, c_f, c_f, c_f, i32, i32, i32, i32, i32, i32, i32, i32, i64, i64, i64, i64, c_f, f32, c_f]),
    "tce_ms_deform_attn_forward_f32": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, i32, i32, i32, i32, i32, c_f]),
    "tce_ms_deform_attn_backward_f32": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, i32, i32, i32, i32, i32, c_f]),
    "tce_msda_fused_f32": (i32, [c_f, c_f, c_f, c_f, C.POINTER(i32), i32, i32, i32, i32, i32, i32, i32, i32, c_f]),
    "tce_msda_fused_valid_f32": (i32, [c_f, c_f, c_f, c_f, C.POINTER(i32), C.POINTER(i32), i32, i32, i32, i32, i32, i32, i32,
                                       i32, c_f]),
    "tce_msda_fewq_raw_f32": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, C.POINTER(i32), C.POINTER(i32), i32, i32, i32, i32, i32, i32, i32,
                                    i32, c_f]),
    "tce_contrastive_f32": (i32, [c_f, c_f, c_f, c_f, i32, i32, i32, i32, c_f]),
    "tce_pos_sine2d_f32": (i32, [c_f, c_f, i32, i32, i32, i32, c_f]),
    "tce_pos_sine2d_valid_f32": (i32, [c_f, c_f, i32, i32, i32, i32, i32, i32, c_f]),
    "tce_resize_nearest_f32": (i32, [c_f, c_f, c_f, i32, i32, i32, i32, i32, i32, c_f]),
    "tce_resize_bilinear_f32": (i32, [c_f, c_f, c_f, i32, i32, i32, i32, i32, i32, c_f]),
    "tce_resize_bilinear_ln_f32": (i32, [c_f, c_f, c_f, c_f, f32, c_f, i32, i32, i32, i32, i32, i32, c_f]),
    "tce_add_f32": (i32, [c_f, c_f, c_f, i64, i64, c_f]),
    "tce_tile_f32": (i32, [c_f, c_f, i64, i64, c_f]),
    "tce_sigmoid_f32": (i32, [c_f, c_f, i64, c_f]),
    "tce_copy_segments": (i32, [C.POINTER(CopySeg), i32, c_f]),
    "tce_box_refine_f32": (i32, [c_f, c_f, c_f, i32, i32, c_f]),
    "tce_mask_pack_f32": (i32, [c_f, c_f, c_f, i32, i32, i32, i32, c_f]),
    "tce_mask_tail_f32": (i32, [c_f, c_f, c_f, i32, c_f, i32, i32, i32, i32, i32, f32, f32, i32, c_f]),
    "tce_select_masks_u8": (i32, [c_f, c_f, c_f, c_f, i32, i32, i32, i32, i32, i32, i32, f32, c_f]),
    "tce_embed_ln_f32": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, f32, i32, c_f]),
    "tce_mha_small64_f32": (i32, [c_f, c_f, i32, i32, f32, c_f]),
    "tce_mha_small64_seqs_f32": (i32, [c_f, i32, c_f, c_f, i32, i32, i32, f32, c_f]),
    "tce_embed_ln_seqs_f32": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, i32, f32, i32, c_f]),
    "tce_tanh_f32": (i32, [c_f, c_f, i64, c_f]),
    "tce_ffn_packed_bytes": (i64, [i32, i32]),
    "tce_ffn_pack_f32": (i32, [c_f, c_f, c_f, c_f, i32, i32, c_f]),
    "tce_ffn_fused_f32": (i32, [c_f, i64, c_f, c_f, c_f, c_f, f32, c_f, c_f, f32, c_f, i64, i32, i32, i32, i32, c_f]),
    "tce_ffn_split_ws_floats": (i64, [i32, i32, i32, i32]),
    "tce_ffn_split_counters": (i32, [i32, i32, i32, i32]),
    "tce_ffn_fused_split_f32": (i32, [c_f, i64, c_f, c_f, c_f, c_f, f32, c_f, c_f, f32, c_f, i64, i32, i32, i32, i32, c_f, i64, c_f, i32, c_f]),
    "tce_xattn_prepare_f32": (i32, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, i32, c_f]),
    "tce_ffn_pack_batched_f32": (i32, [c_f, c_f, c_f, c_f, i32, i32, i32, c_f]),
    "tce_xattn_fused_f32": (i32, [C.POINTER(XattnArgs), c_f]),
    "tce_xattn_ffn_fused_f32": (i32, [C.POINTER(XattnArgs), C.POINTER(XattnFfnArgs), c_f]),
    "tce_ffn_pack_chain_f32": (i32, [c_f, c_f, c_f, c_f, i32, i32, c_f]),
    "tce_xattn_pack_f32": (i32, [c_f, c_f, c_f, c_f, c_f, i32, i32, i32, c_f]),
    "tce_rowlin_packed_bytes": (i64, [i32, i32]),
    "tce_rowlin_pack_f32": (i32, [c_f, i64, c_f, i32, i32, c_f]),
    "tce_rowlin_f32": (i32, [C.POINTER(RowLinArgs), c_f]),
    "tce_conv3x3_packed_bytes": (i64, [i32, i32]),
    "tce_conv3x3_pack_f32": (i32, [c_f, c_f, i32, i32, c_f]),
    "tce_conv3x3_f32": (i32, [c_f, i64, c_f, c_f, c_f, i64, i32, i32, i32, i32, i32, c_f]),
    "tce_fewrow_linear_f32": (i32, [C.POINTER(FewRowArgs), c_f]),
    "tce_thin_linear_splits": (i32, [i32, i32, i32]),
    "tce_thin_partials_f32": (i32, [c_f, i64, i32, c_f, i32, c_f, i64, c_f, i32, i32, i32, c_f]),
    "tce_splitk_reduce_f32": (i32, [c_f, i32, i32, i32, c_f, i32, c_f, i32, i32, c_f, i32, c_f, c_f, f32, c_f]),
    "tce_mha_small64_splits_f32": (i32, [c_f, i32, c_f, c_f, i32, i32, f32, c_f]),
    "tce_swin_attn_packed_bytes": (i64, [i32]),
    "tce_swin_attn_pack_f32": (i32, [c_f, c_f, c_f, i32, c_f]),
    "tce_swin_attn_fused_f32": (i32, [c_f, i64, c_f, c_f, c_f, c_f, c_f, c_f, f32, c_f, i64, i32, i32, i32, i32, i32, c_f]),
    "tce_graph_begin": (i32, [c_f]),
    "tce_graph_end": (i32, [c_f, C.POINTER(C.c_void_p)]),
    "tce_graph_launch": (i32, [C.c_void_p, c_f]),
    "tce_graph_destroy": (i32, [C.c_void_p]),
    "tce_graph_group": (i32, [C.POINTER(C.c_void_p), i32, C.POINTER(C.c_void_p)]),
}

# include/tce_rvos_debug.h: tuning / diagnostic entry points (tools/ only)
DEBUG_SIGNATURES = {
    "tce_gemm_force_tile": (i32, [i32]),
    "tce_debug_set_stamp_buffer": (i32, [c_f]),
    "tce_debug_set_epilogue": (i32, [i32]),
    "tce_debug_ffn_set_stamp_buffer": (i32, [c_f]),
    "tce_debug_ffn_set_half": (i32, [i32]),
    "tce_debug_msda_set_lds": (i32, [i32]),
    "tce_debug_window_attn_set_mfma": (i32, [i32]),
    "tce_debug_mha_set_split": (i32, [i32]),
    "tce_debug_msda_set_fewq": (i32, [i32]),
    "tce_debug_conv3x3_set_waves": (i32, [i32]),
}

_LIB = None


class TceError(RuntimeError):
    pass


def lib():
    """Loads the shared object (once).  Raises if it is missing: there is no CPU / PyTorch fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise TceError(f"{LIB_PATH} not found: build the HIP extension first "
                           f"(python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
        hwq = os.environ.get("GPU_MAX_HW_QUEUES")
        if hwq not in (None, "", "4"):  # measured, profiles/r03_clip_groups.txt: 1-3 abort the HIP runtime at start-up,
            import warnings             # 5-16 double the clip time (every graph edge becomes a cross-queue signal)
            warnings.warn(f"tce_rvos_amd: GPU_MAX_HW_QUEUES={hwq} is set; the HIP runtime's default (4) is the only value this "
                          f"launch program runs well with (1-3 crash the runtime, 5-16 double the clip time)", RuntimeWarning)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
            fn = getattr(l, name)  # AttributeError if the symbol is absent
            fn.restype = res
            fn.argtypes = args
        _LIB = l
    return _LIB


def lib_raw():
    """The CDLL itself even while hazard.recording() has a recording proxy standing in for it."""
    l = lib()
    return l.__dict__.get("_real", l)


def check(status, what=""):
    if status != 0:
        msg = lib().tce_last_error().decode(errors="replace")
        raise TceError(f"{what} failed ({status}): {msg}")
