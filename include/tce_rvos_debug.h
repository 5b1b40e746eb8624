/* tce_rvos_debug.h -- tuning and diagnostic entry points of libtce_rvos.so.  NOT part of the drop-in ABI
 * (include/tce_rvos.h): nothing on the product path calls these; tools/ and the profiling scripts do. */
#ifndef TCE_RVOS_DEBUG_H
#define TCE_RVOS_DEBUG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* tuning aid: force the output tile of the GEMM entry point (0 = automatic; -1 = automatic by the rules of round 4, for A/B) */
int tce_gemm_force_tile(int32_t tile);
/* register (or clear with NULL) a device buffer of >= 2048*8 int64 for the split GEMM's in-kernel s_memtime stamps */
int tce_debug_set_stamp_buffer(long long* dev_buf);
/* tuning aid: 1 (default) LDS-staged coalesced GEMM epilogue stores, 0 direct row-per-lane stores */
int tce_debug_set_epilogue(int32_t lds_staged);
/* register (or clear with NULL) a device buffer of >= 1024*8 int64 for the fused FFN's in-kernel stamps */
int tce_debug_ffn_set_stamp_buffer(long long* dev_buf);
/* tuning aid: the C <= 128 fused MLP as 128-row workgroups, two per CU (same packed stream): 0 = by the launch's round count
 * (default), 1 = always, -1 = never */
int tce_debug_ffn_set_half(int32_t mode);
/* tuning aid: 1 = tce_msda_fused_f32 uses its LDS-staged form for encoder-sized calls (A/B timing, bit-identity
 * tests); 0 = default (measured not faster than the L2-gather form) */
int tce_debug_msda_set_lds(int32_t on); /* 0 default (= 4); 1 LDS-staged; 2 one point at a time; 3 / 4: two / four points in flight */
/* tuning aid: 1 (default) = Swin window attention on the fp32 matrix cores; 0 = the VALU kernel (A/B timing) */
int tce_debug_window_attn_set_mfma(int32_t on);
/* tuning aid: 1 (default) = tce_mha_f32 runs key sequences >= 256 on the fp16 matrix cores (3 x fp16 split) in GEMM
 * modes 1 / 2; 0 = always the exact fp32-MFMA kernel (A/B timing) */
int tce_debug_mha_set_split(int32_t on);
/* tuning aid: 1 (default) = tce_msda_fused_f32 runs calls of <= 8192 (frame, query, head) items with one wavefront per item
 * (all 16 sampling points in flight); 0 = always 8 lanes per item (A/B timing and parity) */
int tce_debug_msda_set_fewq(int32_t on);
/* tuning aid: 0 (default) = tce_conv3x3_f32 picks 128- or 256-pixel workgroups by the rounds of 256 workgroups each form needs;
 * 4 / 8 = always the 4-wave (128-pixel) / 8-wave (256-pixel) form (A/B timing and parity) */
int tce_debug_conv3x3_set_waves(int32_t waves);
#ifdef __cplusplus
}
#endif
#endif
