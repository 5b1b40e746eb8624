/* tce_rvos.h -- C ABI of libtce_rvos.so: the MI355X (gfx950) kernels behind the TCE-RVOS per-clip forward.
 *
 * Conventions ("C-ABI replacement for the native op"):
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated; the caller owns all memory;
 *   - activations are token-major / channels-last: [tokens, C] with tokens = (t, y, x);
 *   - every entry takes the hipStream_t to launch on (as void*), is asynchronous, allocates nothing,
 *     never synchronises, and is therefore legal inside hipGraph capture;
 *   - return value: 0 = launched, <0 = rejected (bad shape / alignment) or launch error; the message is
 *     available from tce_last_error().  No exceptions cross the boundary.
 *
 * Each entry cites the reference code (under /root/reference) whose arithmetic it replaces.
 */
#ifndef TCE_RVOS_H
#define TCE_RVOS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tceStream; /* hipStream_t */

int tce_abi_version(void);
const char* tce_last_error(void);

/* ---------------------------------------------------------------------------------------------------
 * Dense contraction on the fp32 matrix cores (v_mfma_f32_32x32x2_f32):
 *   C[M,N] = epi( (A[M,K] (+ A2[M,K])) @ W[N,K]^T + bias[N] )
 *   epi:  act 0 none | 1 ReLU | 2 GELU(erf);  then res_mode 0 none | 1 "+ res[M,N]" | 2 "* res[M,N]";
 *         act 3 = ReLU applied AFTER the residual (ResNet bottleneck: relu(bn(conv) + identity)).
 * Replaces every nn.Linear / 1x1 Conv2d on the path, and (conv != 0) nn.Conv2d kxk as implicit GEMM over a
 * channels-last image A[T,H,Wd,Cin] with W[N, kh*kw*Cin] (k = (ky*kw+kx)*Cin + c), M = T*Ho*Wo.
 * batch > 1 runs `batch` independent problems (grid.z) with element strides sA.. (0 = shared).
 * Reference: F.linear call sites, e.g. swin_transformer.py:133,151; tce_deformable_transformer.py:489,530;
 * segmentation.py:84-86 (F.conv2d); tce_rvos.py:260,282.
 * Requires K % 16 == 0, lda/lda2/ldw % 4 == 0 and 16-byte aligned A/A2/W.
 */
typedef struct {
  const float* A;
  const float* A2; /* optional addend to A (positional embedding), may be NULL */
  const float* W;
  const float* bias; /* may be NULL */
  const float* res;  /* may be NULL when res_mode == 0 */
  float* C;
  int32_t M, N, K;
  int32_t lda, lda2, ldw, ldc, ldres;
  int32_t act, res_mode;
  int32_t batch;
  int64_t sA, sA2, sW, sBias, sC, sRes;
  int32_t conv; /* 0 plain, 1 implicit-GEMM convolution */
  int32_t T, H, Wd, Cin, Ho, Wo, kh, kw, stride, pad;
} tceGemmArgs;
int tce_gemm_f32(const tceGemmArgs* args, tceStream stream);
/* Same product for skinny, deep problems (M of a few dozen rows, K in the thousands): K is cut into `splits` chunks that
 * run as one batched launch into workspace[splits*M*N], then one pass sums them and applies bias/act/res.  Un-batched
 * GEMMs and (split-fp16 mode) implicit-GEMM convolutions; K % (splits*32) == 0, N % 4 == 0. */
int tce_gemm_splitk_f32(const tceGemmArgs* args, int32_t splits, float* workspace, tceStream stream);
/* The same with the LayerNorm of a post-norm block folded into the reduction pass: C = LayerNorm(A W^T + bias (+ res)) * gamma
 * + beta (act 0, res_mode 0 or 1, N <= 1024, 16-byte aligned rows).  res may alias C (in place on the residual stream).  RoBERTa's attention.output / output sub-layers
 * (transformers' RobertaSelfOutput / RobertaOutput: dense -> dropout -> LayerNorm(hidden + input)), the decoder FFN's
 * linear2 + norm3 (tce_deformable_transformer.py:548-552). */
int tce_gemm_splitk_ln_f32(const tceGemmArgs* args, int32_t splits, float* workspace, const float* gamma, const float* beta,
                           float eps, tceStream stream);
/* which output tile tce_gemm_f32 will use: 128128, 12864 or 6464 (BM*1000-ish code) -- for profiling reports */
int tce_gemm_select_tile(int32_t M, int32_t N, int32_t batch);
int tce_gemm_select_tile_ex(int32_t M, int32_t N, int32_t K, int32_t batch, int32_t conv);
/* arithmetic used by tce_gemm_f32 and the fused kernels: 0 = exact fp32 MFMA (fused kernels off); 1 (default) = fp32
 * operands split on the fly into two fp16 halves, three fp16 MFMAs per product, fp32 accumulation (fp32-accurate to
 * ~3e-7 per product); 2 = ONE fp16 MFMA per product on operands rounded to nearest fp16, fp32 accumulation (BASELINE
 * config 5's "fp16 MFMA"; relative error ~5e-4 per product).  Weight streams packed by tce_ffn_pack_f32 /
 * tce_rowlin_pack_f32 carry the rounding of the mode they were packed in: re-pack after switching. */
int tce_set_gemm_mode(int32_t mode);
int tce_get_gemm_mode(void);
/* Per-THREAD override of the mode (-1 = none: the process default applies).  The mode is read when a launch is issued, so a
 * host thread that switches arithmetic around groups of launches (per-site policies) must not change what another thread's
 * launches -- or captures -- see.  tce_get_gemm_mode() returns the calling thread's effective mode. */
int tce_set_gemm_mode_thread(int32_t mode);

/* Operand-range guard of the split-fp16 arithmetic (its operands must lie inside the fp16 range, |x| < 65504: larger
 * values saturate silently in v_cvt_pkrtz).  Weights are checked on the host when they are packed; activations are
 * checked where they are PRODUCED: every GEMM / fused-FFN epilogue of the split mode sets *flag = 1 when it stores a
 * value with |v| >= 60000 or a NaN.  `flag` is a device int32 owned by the caller (NULL disables the check); the caller
 * reads and clears it at its own synchronisation points (tce_rvos_amd.ops.check_range). */
int tce_set_range_flag(int32_t* flag);

/* LayerNorm over the last dim: out[m,:] = LN(x[m,:] (+ r[m,:])) * gamma + beta.   r may be NULL.
 * Reference: nn.LayerNorm call sites (swin_transformer.py:213,255; tce_deformable_transformer.py:454,...). */
int tce_layernorm_f32(const float* x, const float* r, const float* gamma, const float* beta, float* out,
                      int64_t M, int32_t C, float eps, tceStream stream);

/* GroupNorm on channels-last x[T, HW, C] with G groups of C/G consecutive channels, optional ReLU.
 * ws: device workspace of at least T*G*(nsplit*3 + 2) floats (nsplit = tce_groupnorm_nsplit(HW)).
 * Reference: nn.GroupNorm(32, 256) tce_rvos.py:81,86; nn.GroupNorm(8, C) segmentation.py:43. */
int tce_groupnorm_nsplit(int32_t HW);
int tce_groupnorm_f32(const float* x, const float* gamma, const float* beta, float* out, float* ws, int32_t T,
                      int32_t HW, int32_t C, int32_t G, float eps, int32_t relu, tceStream stream);
/* GroupNorm (+ ReLU) of the coarse map x[T, h, w, C] written through a nearest-neighbour up-sampling onto the finer map and added to it
 * (round 5): out[t, yo, xo, :] = add[t, yo, xo, :] + act(GN(x))[t, yi, xi, :], yi = min(floor(yo * h / ho), h - 1) -- the pixel
 * decoder's top-down merge `cur_fpn + F.interpolate(y, size=cur_fpn.shape[-2:], mode="nearest")` with y = ReLU(GN(conv))
 * (segmentation.py:199-203) as the apply pass of the GroupNorm itself: the normalised coarse map is never written.  Same arithmetic
 * as tce_groupnorm_f32 followed by tce_resize_nearest_f32.  out may alias add (not x); ws as for tce_groupnorm_f32 over h * w. */
int tce_groupnorm_up_add_f32(const float* x, const float* gamma, const float* beta, const float* add, float* out, float* ws,
                             int32_t T, int32_t h, int32_t w, int32_t ho, int32_t wo, int32_t C, int32_t G, float eps,
                             int32_t relu, tceStream stream);

/* ResNet-50 stem (row A11; models/backbone.py:92-96 builds torchvision's resnet50 with FrozenBatchNorm2d :46-56):
 * conv 7x7 stride 2 pad 3, 3 -> 64, + folded frozen BN + ReLU.  frames NCHW [T,3,H,W] -> channels-last
 * [T*Ho*Wo, 64], Ho = (H-1)/2+1.  w_k64 [147][64] with k = (c*7+ky)*7+kx and the BN scale folded in; bias[64]. */
int tce_resnet_stem_f32(const float* frames, const float* w_k64, const float* bias, float* out, int32_t T, int32_t H,
                        int32_t W, tceStream stream);
/* nn.MaxPool2d(3, stride 2, padding 1) of the ResNet stem, channels-last [T,H,W,C] -> [T,(H-1)/2+1,(W-1)/2+1,C]. */
int tce_maxpool3x3s2_cl_f32(const float* x, float* out, int32_t T, int32_t H, int32_t W, int32_t C, tceStream stream);

/* Swin PatchEmbed: frames NCHW [T,3,H,W] -> zero-pad to x4 -> 4x4/s4 conv (w [C,3,4,4]) -> LayerNorm(C).
 * out [T*Hp*Wp, C] token-major.  Reference: swin_transformer.py:427-443. */
int tce_patch_embed_f32(const float* frames, const float* w, const float* b, const float* gamma,
                        const float* beta, float* out, int32_t T, int32_t H, int32_t W, int32_t C, float eps,
                        tceStream stream);

/* Swin (shifted-)window attention core over tokens [T, H, W]: reads packed qkv [T*H*W, 3C] (q|k|v, heads of 32),
 * applies zero-pad-to-x7 (padded tokens carry qkv = bias), cyclic shift, 7x7 windows, q*scale, relative position
 * bias table[169, nH], the -100 shift mask, softmax, AV; writes out [T*H*W, C] in un-shifted token order.
 * Arithmetic: exact-fp32 matrix-core kernel in mode 0; in the split-fp16 / single-pass modes the window runs as the (1,7,7) form
 * of tce_window_attn3d_f32's kernel (fp32-class results, see tce_set_gemm_mode).
 * Reference: swin_transformer.py:50-77,127-158,214-249,370-388. */
int tce_window_attn_f32(const float* qkv, const float* qkv_bias, const float* bias_table, float* out, int32_t T,
                        int32_t H, int32_t W, int32_t C, int32_t nH, int32_t shift, tceStream stream);

/* Video-Swin 3-D (shifted-)window attention core over tokens [T, H, W] (nominal window (8,7,7), shrunk to the
 * grid where it is smaller; temporal/spatial shift (4,3,3) when `shifted` and the dim spans > 1 window; padded
 * tokens carry qkv = bias).  Same operands as tce_window_attn_f32; bias_table is [(2*8-1)*13*13, nH].
 * Reference: video_swin_transformer.py:71-84,138-169,215-249,316-329. */
int tce_window_attn3d_f32(const float* qkv, const float* qkv_bias, const float* bias_table, float* out, int32_t T,
                          int32_t H, int32_t W, int32_t C, int32_t nH, int32_t shifted, tceStream stream);

/* Swin PatchMerging front half: gather 2x2 neighbours in the order (0,0),(1,0),(0,1),(1,1) with odd-size zero
 * padding, LayerNorm(4C).  x [T,H,W,C] -> out [T*ceil(H/2)*ceil(W/2), 4C].  Reference: swin_transformer.py:273-297. */
int tce_patch_merge_ln_f32(const float* x, const float* gamma, const float* beta, float* out, int32_t T, int32_t H,
                           int32_t W, int32_t C, float eps, tceStream stream);

/* Multi-head attention core, head_dim 32, fp32, online softmax:  O = softmax(Q K^T * scale) V per (batch, head).
 * Q [batch][Lq] rows of ldq floats (head h at column h*32), likewise K, V, O.  kmask (optional, uint8
 * [batch, Lk], non-zero = ignore key).  Reference: nn.MultiheadAttention call sites
 * (tce_deformable_transformer.py:467,482,683; segmentation.py:352,366,459). */
int tce_mha_f32(const float* Q, const float* K, const float* V, float* O, int32_t batch, int32_t nheads,
                int32_t Lq, int32_t Lk, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, int64_t sQ, int64_t sK,
                int64_t sV, int64_t sO, const uint8_t* kmask, float scale, tceStream stream);

/* The same attention for LONG key sequences (the pixel decoder's self-attention over a few thousand tokens,
 * segmentation.py:333-361) in GEMM modes 1 / 2: K and V are split to fp16 hi/lo planes once (ws, tce_mha_ws_bytes bytes,
 * 16-byte aligned), the attention waves read their MFMA fragments straight from the planes (no LDS staging, no barrier in
 * the key loop); with few query tiles the 4 waves of a workgroup split the keys.  Same arguments and result as
 * tce_mha_f32 (to fp32 round-off). */
int64_t tce_mha_ws_bytes(int32_t batch, int32_t nheads, int32_t Lk);
int tce_mha_ws_f32(const float* Q, const float* K, const float* V, float* O, void* ws, int32_t batch, int32_t nheads,
                   int32_t Lq, int32_t Lk, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, int64_t sQ, int64_t sK,
                   int64_t sV, int64_t sO, const uint8_t* kmask, float scale, tceStream stream);

/* Multi-scale deformable attention forward -- the drop-in for the reference's only native op
 * MultiScaleDeformableAttention_update.ms_deform_attn_forward (models/ops/src/vision.cpp:13-16,
 * ms_deform_attn_cuda.cu:21-102, ms_deform_im2col_cuda.cuh:34-85,320-455):
 *   value [N,S,M,D], spatial_shapes int64 [L,2] (H,W), level_start_index int64 [L] (both DEVICE),
 *   sampling_loc [N,Lq,M,L,P,2], attn_weight [N,Lq,M,L,P]  ->  out [N,Lq,M*D].
 * Any head dim D and any L, P (as the reference kernel, which is templated over channels).  D == 32 runs the
 * row-gather kernels (one 128-byte value row per corner: 16-byte gathers for L*P <= 16, dword gathers for L*P <= 32);
 * every other shape runs one thread per output scalar. */
int tce_ms_deform_attn_forward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                   const float* sampling_loc, const float* attn_weight, float* out, int32_t N,
                                   int32_t S, int32_t M, int32_t D, int32_t Lq, int32_t L, int32_t P,
                                   tceStream stream);

/* Backward of the same op -- MultiScaleDeformableAttention_update.ms_deform_attn_backward (vision.cpp:13-16,
 * ms_deform_attn_cuda.cu:105-186, ms_deform_im2col_cuda.cuh:87-160,457-1229): grad_output [N,Lq,M*D] ->
 * grad_value [N,S,M,D] (zero-filled here, then accumulated with fp32 atomics: summation order, like the reference's, is
 * not deterministic), grad_sampling_loc [N,Lq,M,L,P,2], grad_attn_weight [N,Lq,M,L,P].  Any D, L, P. */
int tce_ms_deform_attn_backward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                    const float* sampling_loc, const float* attn_weight, const float* grad_output,
                                    float* grad_value, float* grad_sampling_loc, float* grad_attn_weight, int32_t N,
                                    int32_t S, int32_t M, int32_t D, int32_t Lq, int32_t L, int32_t P, tceStream stream);

/* Fused form used by the model: takes the raw projection output proj [N*Lq, M*L*P*3] = (sampling offsets
 * [M,L,P,2] | attention logits [M,L*P]) and the reference points ref [N,Lq,ref_dim] (level-independent,
 * valid_ratios == 1), does softmax over L*P, the offset normalisation (ref_dim 2: / (W_l,H_l); ref_dim 4:
 * / P * wh * 0.5), and the bilinear gather.  Reference: ms_deform_attn.py:98-114 + the native kernel. */
int tce_msda_fused_f32(const float* value, const float* proj, const float* ref, float* out,
                       const int32_t* shapes_hw /* host, [L,2] */, int32_t N, int32_t S, int32_t M, int32_t Lq,
                       int32_t L, int32_t P, int32_t ref_dim, int32_t ref_per_frame, tceStream stream);
/* The same for a padded clip: valid_hw (host, [L,2]) = rows / columns of each level that are NOT padding (rectangular,
 * top-left valid region; NULL = un-padded).  Reference points are multiplied per level by the valid ratios (wv/W, hv/H)
 * (tce_deformable_transformer.py:125-132,180,590-594: get_valid_ratio, reference_points * valid_ratios) and value rows of
 * padded positions read as zero (ms_deform_attn.py:96-97: value.masked_fill(input_padding_mask, 0)). */
int tce_msda_fused_valid_f32(const float* value, const float* proj, const float* ref, float* out,
                             const int32_t* shapes_hw, const int32_t* valid_hw, int32_t N, int32_t S, int32_t M, int32_t Lq,
                             int32_t L, int32_t P, int32_t ref_dim, int32_t ref_per_frame, tceStream stream);

/* Few-query form of the whole attention core WITHOUT the value projection of the frame ("sample, then project"): src [N,S,256] is
 * the module's UN-projected input_flatten, wv [256,256] / bv [256] its value_proj (ms_deform_attn.py:95); out [N*Lq, 256] is what
 * tce_msda_fused_valid_f32 returns on value = value_proj(src) with padded rows zero-filled (:96-97) -- the gather is linear in the
 * value rows, so a head's 32 outputs are its slice of wv applied to the bilinear sample of the RAW rows, plus bias * (sum of the
 * weights of the corners that exist).  For calls with a few dozen queries per frame (frame tokens, decoder queries:
 * tce_deformable_transformer.py:447-454, 688-694) it removes a [N*S, 256] x [256, 256] projection per call.  M must be 8. */
int tce_msda_fewq_raw_f32(const float* src, const float* wv, const float* bv, const float* proj, const float* ref, float* out,
                          const int32_t* shapes_hw, const int32_t* valid_hw, int32_t N, int32_t S, int32_t M, int32_t Lq,
                          int32_t L, int32_t P, int32_t ref_dim, int32_t ref_per_frame, tceStream stream);

/* contrastive_cal (tce_rvos.py:318-319,512-521; --contrastive): out[t] = cosine similarity (eps 1e-6) between the mean over the S
 * rows of frame t of memory [T,S,C] and the sentence feature sent[t / frames_per_clip] ([clips, C]); ws: T*32*C floats. */
int tce_contrastive_f32(const float* memory, const float* sent, float* out, float* ws, int32_t T, int32_t S, int32_t C,
                        int32_t frames_per_clip, tceStream stream);

/* Sine position map of an un-padded [T,h,w] grid, channels-last [T*h*w, 2F] (+ optional per-channel addend,
 * the level embedding).  Reference: position_encoding.py:64-84 (normalize, scale 2*pi, the -0.5 shift). */
int tce_pos_sine2d_f32(float* out, const float* add, int32_t T, int32_t h, int32_t w, int32_t F, tceStream stream);
/* The same for a clip padded to a larger size (position_encoding.py:64-84 with a mask): rows >= hv / columns >= wv of the
 * map are padding (rectangular, top-left valid region: what nested_tensor_from_videos_list produces); the embedding is the
 * cumulative count of non-padded positions normalised by its last value, as the reference computes it. */
int tce_pos_sine2d_valid_f32(float* out, const float* add, int32_t T, int32_t h, int32_t w, int32_t F, int32_t hv, int32_t wv,
                             tceStream stream);

/* Resampling on channels-last maps [T,h,w,C]:
 *   nearest  : out[T,ho,wo,C] = in[.., floor(y*h/ho), floor(x*w/wo), :]  (PyTorch legacy 'nearest'), optional "+ add"
 *   bilinear : align_corners=False, optional "+ add[T,ho,wo,C]"
 * Reference: segmentation.py:212,241 (FPN up), :339-351 (VLBlock down), :360 (VLBlock up). */
int tce_resize_nearest_f32(const float* in, const float* add, float* out, int32_t T, int32_t h, int32_t w, int32_t ho,
                           int32_t wo, int32_t C, tceStream stream);
int tce_resize_bilinear_f32(const float* in, const float* add, float* out, int32_t T, int32_t h, int32_t w,
                            int32_t ho, int32_t wo, int32_t C, tceStream stream);
/* The same followed by a LayerNorm over the C = 256 channels of the sum, as ONE pass (round 5): out = LN(add + bilinear(in)) -- the
 * VisionLanguageBlock's spatially reduced self-attention `tgt + interpolate(attn)` and its norm1 (segmentation.py:357-365).  Same
 * operation sequence as tce_resize_bilinear_f32 followed by tce_layernorm_f32 (equal to a few ulp).  out may alias add (not in). */
int tce_resize_bilinear_ln_f32(const float* in, const float* add, const float* gamma, const float* beta, float eps, float* out,
                               int32_t T, int32_t h, int32_t w, int32_t ho, int32_t wo, int32_t C, tceStream stream);

/* Elementwise helpers (tiny tensors of the decoder / heads):
 *   tce_add_f32        out = a + b (b broadcast with period nb elements)
 *   tce_sigmoid_f32    out = sigmoid(x)
 *   tce_box_refine_f32 out[n,4] = sigmoid(tmp[n,4] + inverse_sigmoid(ref[n,ref_dim]) on the first ref_dim coords)
 *                      (tce_deformable_transformer.py:761-771, util/misc.py:555-559) */
int tce_add_f32(const float* a, const float* b, float* out, int64_t n, int64_t nb, tceStream stream);
/*   tce_tile_f32       out[i] = src[i % n_src] for i < n_src*reps (row broadcast: memory_bus / sentence feature) */
int tce_tile_f32(const float* src, float* out, int64_t n_src, int64_t reps, tceStream stream);
int tce_sigmoid_f32(const float* x, float* out, int64_t n, tceStream stream);

/* Several device-to-device copies in ONE launch: the staging of a captured graph's inputs and the clones of its output
 * tensors (the reference hands back fresh tensors from forward(), tce_rvos.py:360-393; a replayed graph writes into
 * fixed buffers, so each output is copied out -- as one kernel instead of one per tensor).  Sizes in 4-byte words;
 * rows == 1 is a dense run, rows > 1 gathers `rows` runs of row_words from a source pitch into a dense destination. */
#define TCE_COPY_MAX_SEGS 16
typedef struct tceCopySeg {
  const void* src;
  void* dst;
  int64_t rows, row_words, src_pitch_words;
} tceCopySeg;
int tce_copy_segments(const tceCopySeg* segs, int32_t n, tceStream stream);
int tce_box_refine_f32(const float* tmp, const float* ref, float* out, int32_t n, int32_t ref_dim, tceStream stream);

/* Dynamic mask head (tce_rvos.py:426-510,536-599), evaluated for `nl` decoder levels at once without
 * materialising the q-times repeated feature tensor:
 *   tce_mask_pack_f32: params [nl, T*Q, npar] (layout [w0(8x(Cm+2)) | w1(8x8) | w2(1x8) | b0 | b1 | b2])
 *                      -> w0f [T, nl*Q*8, Cm] (first-layer feature weights, GEMM operand) and
 *                         tail [nl, T*Q, 112] (w0x[8], w0y[8], b0[8], w1[64], b1[8], w2[8], b2, pad)
 *   (first layer = tce_gemm_f32 batched over frames: G[T, hw, nl*Q*8] = feats[T, hw, Cm] @ w0f[t]^T)
 *   tce_mask_tail_f32: G + relative coords (ref*(img_w,img_h) - (x*4+2, y*4+2)) -> ReLU -> 8x8 -> ReLU -> 8->1
 *                      -> masks [nl, T, Q, h, w].  G and tail 16-byte aligned. */
int tce_mask_pack_f32(const float* params, float* w0f, float* tail, int32_t nl, int32_t T, int32_t Q, int32_t Cm,
                      tceStream stream);
int tce_mask_tail_f32(const float* G, const float* tail, const float* refs /* [nl, T*Q, ref_ld] */, int32_t ref_ld,
                      float* masks, int32_t nl, int32_t T, int32_t Q, int32_t h, int32_t w, float img_h, float img_w,
                      int32_t stride_px, tceStream stream);

/* Caller harness H (inference_ytvos.py:238-250, inference_davis.py:239-248) as one kernel: best query =
 * argmax_q max_k mean_t sigmoid(pred_logits[t,q,k]); that query's mask logits [T,h,w] are bilinearly up-sampled
 * (align_corners=False) to [T,H0,W0], sigmoid, "> threshold" -> uint8.  best_query (optional) receives the index. */
int tce_select_masks_u8(const float* logits, const float* masks, uint8_t* out, int32_t* best_query, int32_t T,
                        int32_t Q, int32_t K, int32_t h, int32_t w, int32_t H0, int32_t W0, float threshold,
                        tceStream stream);

/* Clip front-end (inference_ytvos.py:38-42,279-287; inference_davis.py:39-43): T.Resize(360) = Pillow
 * Image.resize(BILINEAR) in 8-bit fixed point, then ToTensor + Normalize.  coef [n_out, ksize] int32 (22 fractional
 * bits) and bounds [n_out, 2] = (first input index, taps) are Pillow's precompute_coeffs/normalize_coeffs_8bpc
 * tables (host: tce_rvos_amd/frontend.py).  Pass 1: in [rows, Win, 3] u8 -> tmp [rows, Wout, 3] u8.  Pass 2:
 * tmp [T, Hin, W, 3] u8 -> out [T, 3, Hout, W] f32 through lut[3][256] = ((v/255) - mean[c]) / std[c]. */
int tce_resize_h_u8(const uint8_t* in, const int32_t* coef, const int32_t* bounds, uint8_t* tmp, int64_t rows,
                    int32_t Win, int32_t Wout, int32_t ksize, tceStream stream);
int tce_resize_v_norm_f32(const uint8_t* tmp, const int32_t* coef, const int32_t* bounds, const float* lut, float* out,
                          int32_t T, int32_t Hin, int32_t W, int32_t Hout, int32_t ksize, tceStream stream);

/* RoBERTa text encoder (models/tce_rvos.py:406-424; HuggingFace RobertaModel arithmetic) -- the pieces not covered
 * by tce_gemm_f32 / tce_layernorm_f32:
 *   tce_embed_ln_f32     out[t] = LayerNorm(word[ids[t]] + position[pos_ids[t]] + token_type[0])   (ids int64, device);
 *                        pos_ids NULL: derived in-kernel as HF's create_position_ids_from_input_ids does
 *                        (pad_id + running count of non-pad tokens; pad tokens -> pad_id)
 *   tce_mha_small64_f32  self-attention core over packed qkv [L, 3*nheads*64] (head_dim 64, L <= 128) -> [L, nheads*64]
 *   tce_tanh_f32         pooler activation */
int tce_embed_ln_f32(const int64_t* ids, const int64_t* pos_ids, const float* word, const float* pos, const float* type0,
                     const float* gamma, const float* beta, float* out, int32_t L, int32_t C, float eps, int32_t pad_id,
                     tceStream stream);
/* Several captions of equal length in one launch (clip groups): ids [nseq * seq_len], position ids restart per caption. */
int tce_embed_ln_seqs_f32(const int64_t* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                          const float* beta, float* out, int32_t nseq, int32_t seq_len, int32_t C, float eps, int32_t pad_id,
                          tceStream stream);
int tce_mha_small64_f32(const float* qkv, float* out, int32_t L, int32_t nheads, float scale, tceStream stream);
int tce_tanh_f32(const float* x, float* out, int64_t n, tceStream stream);

/* Thin linear layers as weight STREAMS (csrc/thin.hip): M <= 128 rows against W [N, K] with N % 32 == 0, K % 256 == 0 -- the
 * dense layers of RoBERTa at 32 tokens (HuggingFace RobertaSelfAttention / RobertaSelfOutput / RobertaIntermediate /
 * RobertaOutput under tce_rvos.py:406-424).  One launch requests every byte of W once, from K/256 * N/32 workgroups, in one
 * memory round trip, and leaves K/256 partial planes ws[s][M][N] (s < tce_thin_linear_splits(M, N, K)); the planes meet
 *   - in tce_splitk_reduce_f32:  C = LN?( epi( sum_s ws[s] + bias ) )  (act / res / res_mode as tce_gemm_f32; gamma != NULL:
 *     LayerNorm over N of the sum, as tce_gemm_splitk_ln_f32), or
 *   - in the NEXT consumer's loads: tce_thin_partials_f32 with xsplits > 0 takes its x operand as partial planes
 *     [xsplits][M][K] + bias_x[K] + act_x (0 none, 1 ReLU, 2 GELU(erf)) of the previous layer (fc1 -> fc2 without a
 *     reduction launch); tce_mha_small64_splits_f32 reads the packed qkv projection the same way.
 * Arithmetic: the library's split-fp16 matrix products (GEMM modes 1 / 2; rejected in exact-fp32 mode). */
int32_t tce_thin_linear_splits(int32_t M, int32_t N, int32_t K); /* K/256, or -1 if the shape is not supported */
int tce_thin_partials_f32(const float* x, int64_t ldx, int32_t xsplits, const float* bias_x, int32_t act_x, const float* W,
                          int64_t ldw, float* ws, int32_t M, int32_t N, int32_t K, tceStream stream);
int tce_splitk_reduce_f32(const float* ws, int32_t splits, int32_t M, int32_t N, const float* bias, int32_t act, const float* res,
                          int32_t ldres, int32_t res_mode, float* C, int32_t ldc, const float* gamma, const float* beta, float eps,
                          tceStream stream);
int tce_mha_small64_splits_f32(const float* qkv_planes, int32_t splits, const float* bias, float* out, int32_t L, int32_t nheads,
                               float scale, tceStream stream);
/* The same for nseq sequences of L tokens stacked as [nseq * L, 3E] rows (planes of [nseq * L, 3E]): attention inside each
 * sequence only (several captions of one clip group). */
int tce_mha_small64_seqs_f32(const float* qkv_planes, int32_t splits, const float* bias, float* out, int32_t nseq, int32_t L,
                             int32_t nheads, float scale, tceStream stream);

/* Fused FFN / MLP, the hidden tensor kept on chip (csrc/chain.hip):
 *     out[M,C] = LN_out?( x + W2 act( W1 LN_in?(x) + b1 ) + b2 )        act 1 ReLU | 2 GELU(erf)
 * One launch replaces linear1 -> ReLU -> linear2 -> +residual -> LayerNorm of the transformer / VisionLanguageBlock
 * FFNs (tce_deformable_transformer.py:489-491,548-552; segmentation.py:374-376) and norm2 -> fc1 -> GELU -> fc2 ->
 * +residual of the Swin MLP (swin_transformer.py:28-47,255-256).  g_in/be_in (LayerNorm before W1; the residual is the
 * un-normed x) and g_out/be_out (LayerNorm of the sum) are optional (NULL).  out may alias x.
 * Weights are packed once (static): tce_ffn_pack_f32 takes W1 [Hd,C], b1 [Hd] (may be NULL), W2 [C,Hd] (nn.Linear
 * layouts) and writes tce_ffn_packed_bytes(C,Hd) bytes: fp16 hi/lo planes in MFMA-fragment order.
 * C in {96,128,192,256}, Hd % 32 == 0; row pitches ldx/ldo in floats (% 4 == 0); all pointers 16-byte aligned. */
int64_t tce_ffn_packed_bytes(int32_t C, int32_t Hd);
int tce_ffn_pack_f32(const float* W1, const float* b1, const float* W2, void* packed, int32_t C, int32_t Hd,
                     tceStream stream);
/* `batch` independent weight sets, contiguous ([batch][Hd,C], [batch][Hd], [batch][C,Hd]) -> contiguous streams */
int tce_ffn_pack_batched_f32(const float* W1, const float* b1, const float* W2, void* packed, int32_t C, int32_t Hd,
                             int32_t batch, tceStream stream);
int tce_ffn_fused_f32(const float* x, int64_t ldx, const void* packed, const float* b2, const float* g_in,
                      const float* be_in, float eps_in, const float* g_out, const float* be_out, float eps_out,
                      float* out, int64_t ldo, int32_t M, int32_t C, int32_t Hd, int32_t act, tceStream stream);
/* The same launch with the hidden extent of a row block cut once and the two pieces on different workgroups (round 5): a workgroup
 * owns 128 rows and a launch of 24100 rows is 189 of them -- 0.74 rounds of 256 CUs that cost a full one.  With p blocks handed to
 * p + 1 workgroups a round is ~p/(p+1) as long.  tce_ffn_split_ws_floats > 0 says that a split is planned for the shape (C = 256,
 * ReLU, where the launcher's model says it pays -- csrc/chain.hip ffn_split_plan, calibrated on measurements) and how many floats of workspace it needs; tce_ffn_split_counters how many int32
 * counters.  The counters must be ZERO at the launch and are zero again when it ends (a block's second arriver adds the two partial
 * sums -- the same bits whichever piece arrives last -- runs the epilogue and resets the counter): give each call site that can run
 * concurrently with another its own.  Results differ from tce_ffn_fused_f32's by fp32 round-off (one more addition per element). */
int64_t tce_ffn_split_ws_floats(int32_t M, int32_t C, int32_t Hd, int32_t act);
int32_t tce_ffn_split_counters(int32_t M, int32_t C, int32_t Hd, int32_t act);
int tce_ffn_fused_split_f32(const float* x, int64_t ldx, const void* packed, const float* b2, const float* g_in, const float* be_in,
                            float eps_in, const float* g_out, const float* be_out, float eps_out, float* out, int64_t ldo, int32_t M,
                            int32_t C, int32_t Hd, int32_t act, float* ws, int64_t ws_floats, int32_t* counters, int32_t n_counters,
                            tceStream stream);

/* Token-stationary linear layer (csrc/chain.hip), for K in {96,128,192,256,384,512} and many rows:
 *     out[M,N] = LN_out?( epi( LN_in?(x + a2) W^T + bias ) )      epi: act 0 none | 1 ReLU | 2 GELU(erf), then
 *                                                                 res_mode 0 none | 1 "+ res" | 2 "* res"
 * x stays in registers for the whole launch (read once, in full 128-byte lines), W is streamed from a packed copy
 * (tce_rowlin_pack_f32: fp16 hi/lo planes in MFMA-fragment order; W [N,K] nn.Linear layout, N % 32 == 0).
 * a2: optional addend to x (position map): row pitch lda2; a2_rows > 0 = its row is (row % a2_rows).
 * g_in/be_in: LayerNorm over K applied to (x + a2) before the product; g_out/be_out: LayerNorm over N of the result
 * (N = 256 and K <= 384 only: out_proj + residual + LayerNorm of the post-norm transformer blocks in one launch).
 * batch > 1: grid.y problems with element strides sX, sA2 (0 = shared), sRes, sOut.  out may alias res (not x).
 * Same split-fp16 arithmetic as tce_gemm_f32's default mode.  Reference: the nn.Linear call sites of
 * tce_deformable_transformer.py:439-489,535-548, ops/modules/ms_deform_attn.py:94-101,115, segmentation.py:330-372,
 * swin_transformer.py:133,151. */
typedef struct {
  const float* x;
  const float* a2;
  const void* packed;
  const float* bias;
  const float* res;
  float* out;
  const float *g_in, *be_in, *g_out, *be_out;
  int64_t ldx, lda2, ldres, ldo;
  int64_t sX, sA2, sRes, sOut;
  int32_t M, N, K, batch, a2_rows, act, res_mode;
  float eps_in, eps_out;
} tceRowLinArgs;
int64_t tce_rowlin_packed_bytes(int32_t N, int32_t K);
int tce_rowlin_pack_f32(const float* W, int64_t ldw, void* packed, int32_t N, int32_t K, tceStream stream);
int tce_rowlin_f32(const tceRowLinArgs* args, tceStream stream);

/* 3x3 convolution, stride 1, zero padding 1, 256 -> 256 channels on channels-last maps [T*H*W, 256]: the pixel
 * decoder's output convolutions (CrossModalFPNDecoder, segmentation.py:186-204,253-283; nn.Conv2d(256, 256, 3,
 * padding=1)) as a pixel-stationary kernel (csrc/chain.hip) -- each wave owns 32 pixels and all 256 output channels,
 * the activations go from global memory straight into the MFMA operand registers, only the weights pass through LDS.
 * w [256, 9*256] with k = (ky*3+kx)*256 + c (the layout tce_gemm_f32's implicit-GEMM mode takes); packed streams carry
 * the rounding of the GEMM mode they were packed in (modes 1 and 2; mode 0 has no such kernel: use tce_gemm_f32). */
int64_t tce_conv3x3_packed_bytes(int32_t Cin, int32_t N);
int tce_conv3x3_pack_f32(const float* w, void* packed, int32_t Cin, int32_t N, tceStream stream);
int tce_conv3x3_f32(const float* x, int64_t ldx, const void* packed, const float* bias, float* out, int64_t ldo, int32_t T,
                    int32_t H, int32_t W, int32_t Cin, int32_t N, tceStream stream);

/* Cross-attention of a token tensor against a SHORT key sequence as ONE token-stationary launch (csrc/chain.hip):
 *     out = LN?( res (+|*) ( MHA(q = x + a2, k, v) W_o^T + b_o ) )          8 heads x 32 channels, `group` key slots
 * group 32: the L <= 32 text keys (VisionLanguageBlock / fusion module); group 8: the f_token = 8 frame tokens of
 * FrameTokenLayer's pixel <- token attention (tce_deformable_transformer.py:480-484), keys / values per frame (batch).
 * (VisionLanguageBlock's multihead_attn + residual + norm2, segmentation.py:366-371; VisionLanguageFusionModule,
 * segmentation.py:455-464 with res_mode 2 and no LayerNorm).  With few keys the per-head score is linear in x:
 *     score_h[t,j] = (x+a2)[t,:] . W1[h*32+j,:] + b1[h*32+j],   W1 = scale * k_h W_q,h,  b1 = scale * k_h . b_q,h
 *     out[t,:]     = sum_h softmax_j(score_h[t,:]) . W2[:, h*32+j] + b_o,   W2[:, h*32+j] = W_o,h v_h[j]
 * i.e. linear1 -> (softmax over each group of 32 hidden units) -> linear2: the fused FFN kernel with a grouped softmax
 * as its activation; q, the scores and the per-head outputs never exist in memory.  Per clip:
 *   tce_xattn_prepare_f32  k, v [batch][L,256] (projected keys / values), wqT_ext [257,256] = scale * [W_q^T ; b_q]
 *                          (static), wo [256,256] -> per batch entry W1 [8*group,256], b1 [8*group] (-1e30 for key
 *                          slots >= L), W2 [256,8*group];
 *   tce_ffn_pack_batched_f32 with W1, b1, W2, 256, 8*group, batch -> the weight stream(s);
 *   tce_xattn_fused_f32 -> the launch (sW = bytes between the streams of consecutive batch entries, 0 = shared).
 * a2 / a2_rows / batch strides as in tce_rowlin_f32; res NULL = x; res_mode 1 add | 2 multiply; out may alias x. */
typedef struct {
  const float* x;
  const float* a2;
  const void* packed;
  const float* bo;
  const float* res;
  float* out;
  const float *g_out, *be_out;
  int64_t ldx, lda2, ldres, ldo;
  int64_t sX, sRes, sOut, sW;
  int32_t M, batch, a2_rows, res_mode, group;
  float eps_out;
  int32_t w_div; /* batch entries sharing one weight stream: entry b reads stream b / w_div (0 or 1: one stream per entry when sW != 0).
                    Clip groups: the T frames of a clip share the clip's text keys (VisionLanguageFusionModule per level). */
} tceXattnArgs;
int tce_xattn_prepare_f32(const float* k, const float* v, const float* wqT_ext, const float* wo, float* W1, float* b1,
                          float* W2, int32_t L, int32_t group, int32_t batch, tceStream stream);
int tce_xattn_fused_f32(const tceXattnArgs* args, tceStream stream);

/* Cross-attention -> FFN as ONE launch (round 5): the attention stage exactly as tce_xattn_fused_f32 (its LayerNorm'ed result y is
 * written to `mid`), then, with y still in registers,  out = LN2( y + W2 relu(W1 y + b1) + b2 )  -- multihead_attn + norm2 +
 * linear1/ReLU/linear2 + norm3 of a VisionLanguageBlock (segmentation.py:366-377) or frame_token_atten + norm3 + FFN + norm4 of a
 * FrameTokenLayer (tce_deformable_transformer.py:480-491) without the second launch's row load / the first's row store latency.
 * args->out receives the FFN's result (may alias args->x); `packed` comes from tce_ffn_pack_chain_f32 (same bytes as
 * tce_ffn_pack_f32, W1 in the k order of the accumulator registers); mid [rows, 256] must be a buffer of its own. */
typedef struct {
  const void* packed;  /* tce_ffn_pack_chain_f32(W1 [hidden,256], b1, W2 [256,hidden]) */
  const float* b2;     /* [256] */
  const float* g_out;  /* LayerNorm after the FFN (gamma, beta) or NULL */
  const float* be_out;
  float* mid;          /* [rows, 256] per batch entry: the attention stage's result */
  int64_t ldmid, sMid; /* row pitch / batch stride of mid, in floats */
  int32_t hidden, act; /* FFN width (multiple of 32), activation (1 = ReLU) */
  float eps_out;
} tceXattnFfnArgs;
int tce_ffn_pack_chain_f32(const float* W1, const float* b1, const float* W2, void* packed, int32_t C, int32_t Hd, tceStream stream);
int tce_xattn_ffn_fused_f32(const tceXattnArgs* args, const tceXattnFfnArgs* ffn, tceStream stream);
/* tce_xattn_prepare_f32 + tce_ffn_pack_batched_f32 in ONE launch: k, v [batch][L,256] -> `batch` contiguous weight streams of
 * tce_ffn_packed_bytes(256, 8*group) bytes each, bit-identical to the two-launch form. */
int tce_xattn_pack_f32(const float* k, const float* v, const float* wqT_ext, const float* wo, void* packed, int32_t L, int32_t group,
                       int32_t batch, tceStream stream);

/* Swin attention half-block as ONE launch (csrc/swinattn.hip):
 *     out = x + proj( window_attention( LayerNorm_norm1(x) ) )                                   C in {96, 128, 192, 256}
 * i.e. SwinTransformerBlock.forward up to the first residual (swin_transformer.py:202-249) with WindowAttention.forward
 * (:127-158): norm1, zero padding to a multiple of 7 AFTER the norm (padded tokens carry q, k, v = bias), cyclic shift,
 * 7x7 window partition, packed qkv projection, q * 32^-0.5, relative position bias table [169, C/32], the -100 shift mask
 * of the padded grid (:370-388), softmax, P v, output projection, window reverse / un-shift / crop, + x.  x, out: tokens
 * [T*H*W, C] with row pitches ldx / ldo; out may BE x (in place: the Swin residual stream) or must not overlap it.
 * Replaces tce_rowlin_f32 (norm1 -> qkv) + tce_window_attn_f32 + tce_gemm_f32 (proj + residual) and the [tokens, 3C] qkv
 * tensor between them.  Weights are packed once (tce_swin_attn_pack_f32: Wqkv [3C, C], Wproj [C, C] in nn.Linear layout ->
 * tce_swin_attn_packed_bytes(C) bytes: fp16 hi/lo planes in the order the kernel consumes them, carrying the rounding of
 * the GEMM mode they were packed in).  Arithmetic: the library's split-fp16 matrix products (modes 1 and 2). */
int64_t tce_swin_attn_packed_bytes(int32_t C);
int tce_swin_attn_pack_f32(const float* Wqkv, const float* Wproj, void* packed, int32_t C, tceStream stream);
int tce_swin_attn_fused_f32(const float* x, int64_t ldx, const void* packed, const float* qkv_bias, const float* proj_bias,
                            const float* bias_table, const float* gamma1, const float* beta1, float eps, float* out, int64_t ldo,
                            int32_t T, int32_t H, int32_t W, int32_t C, int32_t shift, tceStream stream);

/* hipGraph helpers so that the Python host can capture one forward and replay it. */
int tce_graph_begin(tceStream stream);
int tce_graph_end(tceStream stream, void** graph_exec_out);
int tce_graph_launch(void* graph_exec, tceStream stream);
int tce_graph_destroy(void* graph_exec);
/* graphs = n hipGraph_t handles (captured, NOT instantiated; e.g. torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()): builds ONE
 * executable in which they are n components with no edge between them -- every kernel / memset / empty node is re-created with
 * its parameters and edges in a fresh graph, which is then instantiated (launch: tce_graph_launch; free: tce_graph_destroy; the
 * inputs are left untouched and may be destroyed afterwards).  With n = 1 this is how the host builds the executable of a clip
 * (round 5): an executable made of explicitly added nodes keeps no reference to the capture's streams, so -- unlike one
 * instantiated from the captured graph itself (a HIP runtime defect) -- it can be destroyed while other
 * executables live.  Returns an error (and leaves *graph_exec_out alone) for node kinds it cannot re-create (memcpy, host,
 * child-graph nodes: none in the shipped launch program).  (Round 3 used it with n = 2 to run two clips as one graph:
 * bit-identical, not faster -- DESIGN.md section 3.8; TCE_GROUP_CHILD=1 in the environment: child-graph nodes instead.) */
int tce_graph_group(void* const* graphs, int32_t n, void** graph_exec_out);

/* ---------------------------------------------------------------------------------------------------
 * Few-row linear layers (R of a few dozen rows; csrc/fewrow.hip): up to three projections of the SAME rows in one launch,
 *     out_s[r, n] = act_s( sum_k (x[r,k] (+ a2[r % a2_rows, k] if seg.use_a2)) W_s[n,k] + bias_s[n] )  (+ res[r, n] for s = 0),
 * exact fp32 on the vector ALUs.  act: 0 none, 1 ReLU, 2 sigmoid, 3 GELU (erf).  a2_rows = 0: the addend has one row per x
 * row.  `out` of a segment must not overlap x or a2 (rows are re-read by other workgroups), the outputs of different segments
 * must not overlap each other, res must not overlap the outputs of segments 1 / 2; out_0 may alias res exactly (in place on
 * the residual stream: same pointer, same pitch).  ldx >= K.
 * Replaces the per-token nn.Linear / sigmoid call sites of the frame-token layer, the decoder's per-query projections and
 * the text-side key / value projections (tce_deformable_transformer.py:439-484,665-790; segmentation.py:366-371).
 */
typedef struct {
  const float* W;    /* [N, K], row pitch ldw (multiple of 4 floats, 16-byte aligned) */
  const float* bias; /* [N] or NULL */
  float* out;        /* [R, N], row pitch ldo */
  int32_t N, ldw, ldo;
  int32_t use_a2; /* add a2 to x for this projection */
  int32_t act;
} tceFewRowSeg;
typedef struct {
  const float* x;   /* [R, K], row pitch ldx */
  const float* a2;  /* optional addend rows, pitch lda2 */
  const float* res; /* optional residual added to segment 0, pitch ldres */
  int64_t ldx, lda2, ldres;
  int32_t a2_rows, R, K, nseg;
  tceFewRowSeg seg[3];
  /* round 5 (ABI 5), optional LayerNorm PROLOGUE (K <= 256): x rows are normalised over K (gamma g_in, beta be_in, eps_in) before
   * the addend and the projections -- the post-norm LayerNorm in front of a few-row projection (tce_deformable_transformer.py:
   * 454-455,468-469) rides in the projection's launch.  xn_out (optional, [R, K], pitch ldxn, must not overlap x / a2 / any out):
   * the normalised rows are also written there (by the workgroups of output slab 0), for the residual stream. */
  const float* g_in;
  const float* be_in;
  float* xn_out;
  int64_t ldxn;
  float eps_in;
} tceFewRowArgs;
int tce_fewrow_linear_f32(const tceFewRowArgs* args, tceStream stream);

#ifdef __cplusplus
}
#endif
#endif /* TCE_RVOS_H */
