// Small HBM-bound kernels: sine position maps, nearest / bilinear resampling on channels-last maps,
// elementwise helpers of the decoder, and the dynamic mask head's pack + tail stages.
#include "common.h"
#include "../../include/tce_rvos.h"

namespace {

// position_encoding.py:64-84 for an all-valid mask: y_embed = (y+1-0.5)/(h+1e-6)*2pi, channels [pos_y(F) | pos_x(F)],
// channel i: embed / 10000^(2*(i/2)/F), sin for even i, cos for odd i.
// Padded clips (rows >= hv / columns >= wv are padding; rectangular masks only): y_embed is the cumulative count of
// non-padded rows IN THIS COLUMN -- min(y + 1, hv) in a valid column, 0 in a padded one -- normalised by its last value.
__global__ void __launch_bounds__(256) pos_sine2d_kernel(float* __restrict__ out, const float* __restrict__ add, int T,
                                                         int h, int w, int F, long long total, int hv, int wv) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int C = 2 * F;
  const int c = (int)(idx % C);
  const long long tok = idx / C;
  const int x = (int)(tok % w);
  const int y = (int)((tok / w) % h);
  const float two_pi = 6.283185307179586f;
  const bool is_y = c < F;
  const int i = is_y ? c : c - F;
  const float ye = x < wv ? (float)min(y + 1, hv) : 0.f, yl = x < wv ? (float)hv : 0.f;  // cumsum value here / in the last row
  const float xe = y < hv ? (float)min(x + 1, wv) : 0.f, xl = y < hv ? (float)wv : 0.f;
  const float e = is_y ? (ye - 0.5f) / (yl + 1e-6f) * two_pi : (xe - 0.5f) / (xl + 1e-6f) * two_pi;
  const float dim_t = powf(10000.0f, (float)(2 * (i / 2)) / (float)F);
  const float v = e / dim_t;
  float r = (i & 1) ? cosf(v) : sinf(v);
  if (add) r += add[c];
  out[idx] = r;
}

__global__ void __launch_bounds__(256) resize_nearest_kernel(const float* __restrict__ in, const float* __restrict__ add,
                                                             float* __restrict__ out, int T, int h, int w, int ho,
                                                             int wo, int C4, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C4);
  long long r = idx / C4;
  const int xo = (int)(r % wo); r /= wo;
  const int yo = (int)(r % ho);
  const int t = (int)(r / ho);
  const float sy = (float)h / (float)ho, sx = (float)w / (float)wo;
  const int yi = min((int)floorf((float)yo * sy), h - 1);
  const int xi = min((int)floorf((float)xo * sx), w - 1);
  f32x4 v = reinterpret_cast<const f32x4*>(in)[(((long long)t * h + yi) * w + xi) * C4 + c];
  if (add) v += reinterpret_cast<const f32x4*>(add)[idx];
  reinterpret_cast<f32x4*>(out)[idx] = v;
}

// four channels of output pixel (t, yo, xo) of F.interpolate(mode="bilinear", align_corners=False); one fixed sequence of
// operations (no contraction left to the compiler) shared by the plain kernel and the fused resize + add + LayerNorm kernel, so the
// two give the same bits
__device__ __forceinline__ f32x4 bilinear4(const float* __restrict__ in, const int t, const int yo, const int xo, const int c,
                                           const int h, const int w, const int ho, const int wo, const int C4) {
#pragma clang fp contract(off)
  const float sy = (float)h / (float)ho, sx = (float)w / (float)wo;
  const float fy = fmaxf(sy * ((float)yo + 0.5f) - 0.5f, 0.f);
  const float fx = fmaxf(sx * ((float)xo + 0.5f) - 0.5f, 0.f);
  const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const f32x4* i4 = reinterpret_cast<const f32x4*>(in) + (long long)t * h * w * C4 + c;
  const f32x4 v00 = i4[((long long)y0 * w + x0) * C4], v01 = i4[((long long)y0 * w + x1) * C4];
  const f32x4 v10 = i4[((long long)y1 * w + x0) * C4], v11 = i4[((long long)y1 * w + x1) * C4];
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float top = __builtin_fmaf(lx, v01[j], hx * v00[j]), bot = __builtin_fmaf(lx, v11[j], hx * v10[j]);
    o[j] = __builtin_fmaf(ly, bot, hy * top);
  }
  return o;
}

__global__ void __launch_bounds__(256) resize_bilinear_kernel(const float* __restrict__ in,
                                                              const float* __restrict__ add, float* __restrict__ out,
                                                              int T, int h, int w, int ho, int wo, int C4,
                                                              long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C4);
  long long r = idx / C4;
  const int xo = (int)(r % wo); r /= wo;
  const int yo = (int)(r % ho);
  const int t = (int)(r / ho);
  f32x4 o = bilinear4(in, t, yo, xo, c, h, w, ho, wo, C4);
  if (add) o += reinterpret_cast<const f32x4*>(add)[idx];
  reinterpret_cast<f32x4*>(out)[idx] = o;
}

// out[row, :] = LayerNorm(add[row, :] + bilinear(in)[row, :]) for C = 256: a wave per output pixel (64 lanes x 4 channels), the
// statistics by xor-shuffles exactly as layernorm_reg_kernel<64> takes them -- the VisionLanguageBlock's spatially reduced
// self-attention `tgt + interpolate(attention output)` followed by norm1 (segmentation.py:357-365) as ONE pass over the map (the two
// launches read and write the [T*h*w, 256] map twice: 590 MB each way at stride 4 for an 8-clip group).  out may alias add.
__global__ void __launch_bounds__(256) resize_bilinear_ln_kernel(const float* __restrict__ in, const float* add,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* out, int T, int h, int w, int ho, int wo, float eps,
                                                                 long long rows) {
  constexpr int C = 256, C4 = 64;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;  // (whole waves: the shuffles below stay inside a wave)
  long long r = row;
  const int xo = (int)(r % wo); r /= wo;
  const int yo = (int)(r % ho);
  const int t = (int)(r / ho);
  f32x4 v = bilinear4(in, t, yo, xo, lane, h, w, ho, wo, C4);
  v += reinterpret_cast<const f32x4*>(add + row * C)[lane];
  float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  const float a = v[0] - mean, b = v[1] - mean, c = v[2] - mean, d = v[3] - mean;
  float q = (a * a + b * b) + (c * c + d * d);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)C + eps);
  const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[lane], bb = reinterpret_cast<const f32x4*>(beta)[lane];
  f32x4 o;
  o[0] = a * rstd * g[0] + bb[0];
  o[1] = b * rstd * g[1] + bb[1];
  o[2] = c * rstd * g[2] + bb[2];
  o[3] = d * rstd * g[3] + bb[3];
  reinterpret_cast<f32x4*>(out + row * C)[lane] = o;
}

__global__ void __launch_bounds__(256) add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long long n, long long nb) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i % nb];
}

__global__ void __launch_bounds__(256) tile_kernel(const float* __restrict__ src, float* __restrict__ out, long long n,
                                                   long long n_src) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = src[i % n_src];
}

__global__ void __launch_bounds__(256) sigmoid_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                      long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = 1.f / (1.f + expf(-x[i]));
}

__device__ __forceinline__ float inv_sigmoid(float x) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float x1 = fmaxf(x, 1e-5f), x2 = fmaxf(1.f - x, 1e-5f);
  return logf(x1 / x2);
}

__global__ void __launch_bounds__(256) box_refine_kernel(const float* __restrict__ tmp, const float* __restrict__ ref,
                                                         float* __restrict__ out, int n, int ref_dim) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * 4) return;
  const int r = i >> 2, c = i & 3;
  float v = tmp[i];
  if (c < ref_dim) v += inv_sigmoid(ref[r * ref_dim + c]);
  out[i] = 1.f / (1.f + expf(-v));
}

// ---- caller harness H (inference_ytvos.py:238-250): best query, bilinear up-sampling, sigmoid, threshold ----
__global__ void __launch_bounds__(256) harness_kernel(const float* __restrict__ logits, const float* __restrict__ masks,
                                                      uint8_t* __restrict__ out, int* __restrict__ best_out, int T,
                                                      int Q, int K, int h, int w, int H0, int W0, float threshold) {
  __shared__ int s_best;
  if (threadIdx.x == 0) {
    // pred_scores = sigmoid(logits).mean(frames); max over classes; argmax over queries (first maximum wins)
    float best = -1.f;
    int bq = 0;
    for (int q = 0; q < Q; ++q) {
      float mx = -1.f;
      for (int k = 0; k < K; ++k) {
        float sum = 0.f;
        for (int t = 0; t < T; ++t) sum += 1.f / (1.f + expf(-logits[((long long)t * Q + q) * K + k]));
        mx = fmaxf(mx, sum / (float)T);
      }
      if (mx > best) {
        best = mx;
        bq = q;
      }
    }
    s_best = bq;
    if (blockIdx.x == 0 && best_out) *best_out = bq;
  }
  __syncthreads();
  const int bq = s_best;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)T * H0 * W0;
  if (idx >= total) return;
  const int xo = (int)(idx % W0);
  const int yo = (int)((idx / W0) % H0);
  const int t = (int)(idx / ((long long)W0 * H0));
  const float sy = (float)h / (float)H0, sx = (float)w / (float)W0;
  const float fy = fmaxf(sy * ((float)yo + 0.5f) - 0.5f, 0.f);
  const float fx = fmaxf(sx * ((float)xo + 0.5f) - 0.5f, 0.f);
  const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float* m = masks + ((long long)t * Q + bq) * h * w;
  const float v = hy * (hx * m[y0 * w + x0] + lx * m[y0 * w + x1]) + ly * (hx * m[y1 * w + x0] + lx * m[y1 * w + x1]);
  out[idx] = (1.f / (1.f + expf(-v)) > threshold) ? 1 : 0;
}

// ---- dynamic mask head -------------------------------------------------------------------------------
constexpr int DC = 8;        // dynamic_mask_channels
constexpr int TAIL_LD = 112;  // floats per (level, frame, query) in the packed tail parameter block

__global__ void __launch_bounds__(256) mask_pack_kernel(const float* __restrict__ params, float* __restrict__ w0f,
                                                        float* __restrict__ tail, int nl, int T, int Q, int Cm) {
  // one workgroup per (level, frame, query)
  const int item = blockIdx.x;  // (lvl*T + t)*Q + q
  const int q = item % Q, t = (item / Q) % T, lvl = item / (Q * T);
  const int npar = DC * (Cm + 2) + DC * DC + DC + DC + DC + 1;
  const float* p = params + ((long long)lvl * T * Q + t * Q + q) * npar;
  float* wrow = w0f + ((long long)t * (nl * Q * DC) + (lvl * Q + q) * DC) * Cm;
  for (int i = threadIdx.x; i < DC * Cm; i += 256) {
    const int c = i / Cm, k = i - c * Cm;
    wrow[(long long)c * Cm + k] = p[c * (Cm + 2) + k];
  }
  float* tl = tail + ((long long)lvl * T * Q + t * Q + q) * TAIL_LD;
  const int off1 = DC * (Cm + 2), off2 = off1 + DC * DC, off3 = off2 + DC;
  for (int i = threadIdx.x; i < TAIL_LD; i += 256) {
    float v = 0.f;
    if (i < 8) v = p[i * (Cm + 2) + Cm];                  // w0 x-coordinate column
    else if (i < 16) v = p[(i - 8) * (Cm + 2) + Cm + 1];  // w0 y-coordinate column
    else if (i < 24) v = p[off3 + (i - 16)];              // b0
    else if (i < 88) v = p[off1 + (i - 24)];              // w1 [out][in]
    else if (i < 96) v = p[off3 + 8 + (i - 88)];          // b1
    else if (i < 104) v = p[off2 + (i - 96)];             // w2
    else if (i == 104) v = p[off3 + 16];                  // b2
    tl[i] = v;
  }
}

// Pixel-stationary: a thread owns one pixel of frame t and walks a chunk of the frame's (level, query) items, whose packed
// tails sit in LDS (16-byte broadcast reads).  The pixel's G row ([nl*Q*DC] floats, contiguous) is read once by its own
// thread, so every 128-byte line of G comes from HBM once -- the first form (one launch row per item, each reading 32 of a
// line's 128 bytes) fetched G four times over (190 MB against 52 MB algorithmic, profiles/r02_pmc_traffic.json).  (Reading
// the wave-uniform tails through the scalar cache instead of LDS was tried: 45 us against 24 -- one s_load round trip per item.)
constexpr int TAIL_ITEMS = 8;  // items per workgroup = a thread's serial chain (config 2: 20.8 us with all 20 items in one
                               // thread, 16.0 us at 8; the first form 28.3 us); the chunks of one pixel block are neighbours
                               // in blockIdx.x, so they meet G's lines in cache
__global__ void __launch_bounds__(128) mask_tail_kernel(const float* __restrict__ G, const float* __restrict__ tail,
                                                        const float* __restrict__ refs, int ref_ld,
                                                        float* __restrict__ masks, int nl, int T, int Q, int h, int w,
                                                        float img_h, float img_w, int stride_px) {
  __shared__ __attribute__((aligned(16))) float sp[TAIL_ITEMS * TAIL_LD];
  __shared__ float sref[TAIL_ITEMS * 2];
  const int t = blockIdx.y;
  const int nitem = nl * Q;
  const int nchunk = (nitem + TAIL_ITEMS - 1) / TAIL_ITEMS;
  const int chunk = blockIdx.x % nchunk, pblock = blockIdx.x / nchunk;
  const int it0 = chunk * TAIL_ITEMS, itn = min(TAIL_ITEMS, nitem - it0);
  for (int i = threadIdx.x; i < itn * (TAIL_LD / 4); i += 128) {
    const int it = it0 + i / (TAIL_LD / 4), lvl = it / Q, q = it - lvl * Q;
    reinterpret_cast<f32x4*>(sp)[i] =
        reinterpret_cast<const f32x4*>(tail + ((long long)lvl * T * Q + t * Q + q) * TAIL_LD)[i % (TAIL_LD / 4)];
  }
  for (int i = threadIdx.x; i < itn * 2; i += 128) {
    const int it = it0 + (i >> 1), lvl = it / Q, q = it - lvl * Q;
    sref[i] = refs[((long long)lvl * T * Q + t * Q + q) * ref_ld + (i & 1)];
  }
  __syncthreads();
  const int pix = pblock * 128 + threadIdx.x;
  const int hw = h * w;
  if (pix >= hw) return;
  const int y = pix / w, x = pix - y * w;
  const float px = (float)(x * stride_px + stride_px / 2), py = (float)(y * stride_px + stride_px / 2);
  const float* g = G + ((long long)t * hw + pix) * (nitem * DC) + (long long)it0 * DC;
#pragma unroll 2
  for (int i = 0; i < itn; ++i) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(sp + i * TAIL_LD);
    float s[TAIL_LD - 4];  // 108 floats: the 105 used + padding
#pragma unroll
    for (int j = 0; j < (TAIL_LD - 4) / 4; ++j) {
      const f32x4 v = s4[j];
      s[4 * j] = v[0]; s[4 * j + 1] = v[1]; s[4 * j + 2] = v[2]; s[4 * j + 3] = v[3];
    }
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(g + i * DC), g1 = *reinterpret_cast<const f32x4*>(g + i * DC + 4);
    const float relx = sref[2 * i] * img_w - px;
    const float rely = sref[2 * i + 1] * img_h - py;
    float h0[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) {
      const float gv = c < 4 ? g0[c] : g1[c - 4];
      h0[c] = fmaxf(gv + s[c] * relx + s[8 + c] * rely + s[16 + c], 0.f);
    }
    float o = s[104];
#pragma unroll
    for (int c = 0; c < DC; ++c) {
      float a = s[88 + c];
#pragma unroll
      for (int k = 0; k < DC; ++k) a = fmaf(s[24 + c * DC + k], h0[k], a);
      o = fmaf(s[96 + c], fmaxf(a, 0.f), o);
    }
    const int it = it0 + i, lvl = it / Q, q = it - lvl * Q;
    masks[(((long long)lvl * T + t) * Q + q) * hw + pix] = o;
  }
}

// One launch for a list of copies (the graph's input staging and the clones of its output tensors): blockIdx.y picks the
// segment, blockIdx.x strides over it.  Dense 16-byte-aligned segments move as float4; anything else word by word.
struct CopySegs {
  tceCopySeg s[TCE_COPY_MAX_SEGS];
};
__global__ void __launch_bounds__(256) copy_segments_kernel(CopySegs segs) {
  const tceCopySeg sg = segs.s[blockIdx.y];
  const long long total = sg.rows * sg.row_words;
  const long long stride = (long long)gridDim.x * 256;
  const float* __restrict__ src = (const float*)sg.src;
  float* __restrict__ dst = (float*)sg.dst;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (sg.rows == 1 && (((uintptr_t)sg.src | (uintptr_t)sg.dst) & 15) == 0) {
    const long long quads = total >> 2;
    for (long long q = i; q < quads; q += stride) ((float4*)dst)[q] = ((const float4*)src)[q];
    for (long long t = (quads << 2) + i; t < total; t += stride) dst[t] = src[t];
    return;
  }
  for (; i < total; i += stride) {
    const long long r = i / sg.row_words, c = i - r * sg.row_words;
    dst[i] = src[r * sg.src_pitch_words + c];
  }
}


}  // namespace

extern "C" int tce_pos_sine2d_f32(float* out, const float* add, int32_t T, int32_t h, int32_t w, int32_t F,
                                  tceStream stream) {
  return tce_pos_sine2d_valid_f32(out, add, T, h, w, F, h, w, stream);
}

extern "C" int tce_pos_sine2d_valid_f32(float* out, const float* add, int32_t T, int32_t h, int32_t w, int32_t F, int32_t hv,
                                        int32_t wv, tceStream stream) {
  TCE_CHECK_ARG(out && T > 0 && h > 0 && w > 0 && F > 0 && hv >= 1 && hv <= h && wv >= 1 && wv <= w,
                "tce_pos_sine2d_f32: bad arguments");
  const long long total = (long long)T * h * w * 2 * F;
  hipLaunchKernelGGL(pos_sine2d_kernel, dim3(tce_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, out, add, T, h, w,
                     F, total, hv, wv);
  TCE_CHECK_LAUNCH("tce_pos_sine2d_f32");
  return TCE_OK;
}

extern "C" int tce_resize_nearest_f32(const float* in, const float* add, float* out, int32_t T, int32_t h, int32_t w,
                                      int32_t ho, int32_t wo, int32_t C, tceStream stream) {
  TCE_CHECK_ARG(in && out && T > 0 && h > 0 && w > 0 && ho > 0 && wo > 0 && C > 0 && C % 4 == 0,
                "tce_resize_nearest_f32: bad arguments");
  TCE_CHECK_ARG(tce_aligned16(in) && tce_aligned16(out) && (!add || tce_aligned16(add)),
                "tce_resize_nearest_f32: pointers must be 16-byte aligned");
  const long long total = (long long)T * ho * wo * (C / 4);
  hipLaunchKernelGGL(resize_nearest_kernel, dim3(tce_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, add, out,
                     T, h, w, ho, wo, C / 4, total);
  TCE_CHECK_LAUNCH("tce_resize_nearest_f32");
  return TCE_OK;
}

extern "C" int tce_resize_bilinear_f32(const float* in, const float* add, float* out, int32_t T, int32_t h, int32_t w,
                                       int32_t ho, int32_t wo, int32_t C, tceStream stream) {
  TCE_CHECK_ARG(in && out && T > 0 && h > 0 && w > 0 && ho > 0 && wo > 0 && C > 0 && C % 4 == 0,
                "tce_resize_bilinear_f32: bad arguments");
  TCE_CHECK_ARG(tce_aligned16(in) && tce_aligned16(out) && (!add || tce_aligned16(add)),
                "tce_resize_bilinear_f32: pointers must be 16-byte aligned");
  const long long total = (long long)T * ho * wo * (C / 4);
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(tce_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, add,
                     out, T, h, w, ho, wo, C / 4, total);
  TCE_CHECK_LAUNCH("tce_resize_bilinear_f32");
  return TCE_OK;
}

extern "C" int tce_resize_bilinear_ln_f32(const float* in, const float* add, const float* gamma, const float* beta, float eps,
                                          float* out, int32_t T, int32_t h, int32_t w, int32_t ho, int32_t wo, int32_t C,
                                          tceStream stream) {
  TCE_CHECK_ARG(in && add && gamma && beta && out && T > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "tce_resize_bilinear_ln_f32: bad arguments");
  TCE_CHECK_ARG(C == 256, "tce_resize_bilinear_ln_f32: C must be 256 (a wave per row)");
  TCE_CHECK_ARG(tce_aligned16(in) && tce_aligned16(out) && tce_aligned16(add) && tce_aligned16(gamma) && tce_aligned16(beta),
                "tce_resize_bilinear_ln_f32: pointers must be 16-byte aligned");
  TCE_CHECK_ARG(in != out, "tce_resize_bilinear_ln_f32: out may alias add, not in");
  const long long rows = (long long)T * ho * wo;
  hipLaunchKernelGGL(resize_bilinear_ln_kernel, dim3(tce_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, in, add, gamma, beta, out,
                     T, h, w, ho, wo, eps, rows);
  TCE_CHECK_LAUNCH("tce_resize_bilinear_ln_f32");
  return TCE_OK;
}

extern "C" int tce_add_f32(const float* a, const float* b, float* out, int64_t n, int64_t nb, tceStream stream) {
  TCE_CHECK_ARG(a && b && out && n > 0 && nb > 0, "tce_add_f32: bad arguments");
  hipLaunchKernelGGL(add_kernel, dim3(tce_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long long)n,
                     (long long)nb);
  TCE_CHECK_LAUNCH("tce_add_f32");
  return TCE_OK;
}

extern "C" int tce_tile_f32(const float* src, float* out, int64_t n_src, int64_t reps, tceStream stream) {
  TCE_CHECK_ARG(src && out && n_src > 0 && reps > 0, "tce_tile_f32: bad arguments");
  const long long n = (long long)n_src * reps;
  hipLaunchKernelGGL(tile_kernel, dim3(tce_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, src, out, n,
                     (long long)n_src);
  TCE_CHECK_LAUNCH("tce_tile_f32");
  return TCE_OK;
}

extern "C" int tce_sigmoid_f32(const float* x, float* out, int64_t n, tceStream stream) {
  TCE_CHECK_ARG(x && out && n > 0, "tce_sigmoid_f32: bad arguments");
  hipLaunchKernelGGL(sigmoid_kernel, dim3(tce_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, (long long)n);
  TCE_CHECK_LAUNCH("tce_sigmoid_f32");
  return TCE_OK;
}

extern "C" int tce_box_refine_f32(const float* tmp, const float* ref, float* out, int32_t n, int32_t ref_dim,
                                  tceStream stream) {
  TCE_CHECK_ARG(tmp && ref && out && n > 0 && (ref_dim == 2 || ref_dim == 4), "tce_box_refine_f32: bad arguments");
  hipLaunchKernelGGL(box_refine_kernel, dim3(tce_cdiv(n * 4, 256)), dim3(256), 0, (hipStream_t)stream, tmp, ref, out, n,
                     ref_dim);
  TCE_CHECK_LAUNCH("tce_box_refine_f32");
  return TCE_OK;
}

extern "C" int tce_select_masks_u8(const float* logits, const float* masks, uint8_t* out, int32_t* best_query,
                                   int32_t T, int32_t Q, int32_t K, int32_t h, int32_t w, int32_t H0, int32_t W0,
                                   float threshold, tceStream stream) {
  TCE_CHECK_ARG(logits && masks && out && T > 0 && Q > 0 && K > 0 && h > 0 && w > 0 && H0 > 0 && W0 > 0,
                "tce_select_masks_u8: bad arguments");
  const long long total = (long long)T * H0 * W0;
  hipLaunchKernelGGL(harness_kernel, dim3(tce_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, logits, masks, out,
                     best_query, T, Q, K, h, w, H0, W0, threshold);
  TCE_CHECK_LAUNCH("tce_select_masks_u8");
  return TCE_OK;
}

extern "C" int tce_mask_pack_f32(const float* params, float* w0f, float* tail, int32_t nl, int32_t T, int32_t Q,
                                 int32_t Cm, tceStream stream) {
  TCE_CHECK_ARG(params && w0f && tail && nl > 0 && T > 0 && Q > 0 && Cm > 0 && Cm % 16 == 0,
                "tce_mask_pack_f32: bad arguments");
  hipLaunchKernelGGL(mask_pack_kernel, dim3(nl * T * Q), dim3(256), 0, (hipStream_t)stream, params, w0f, tail, nl, T, Q,
                     Cm);
  TCE_CHECK_LAUNCH("tce_mask_pack_f32");
  return TCE_OK;
}

extern "C" int tce_mask_tail_f32(const float* G, const float* tail, const float* refs, int32_t ref_ld, float* masks,
                                 int32_t nl, int32_t T, int32_t Q, int32_t h, int32_t w, float img_h, float img_w,
                                 int32_t stride_px, tceStream stream) {
  TCE_CHECK_ARG(G && tail && refs && masks && nl > 0 && T > 0 && Q > 0 && h > 0 && w > 0 && ref_ld >= 2,
                "tce_mask_tail_f32: bad arguments");
  TCE_CHECK_ARG(tce_aligned16(G) && tce_aligned16(tail), "tce_mask_tail_f32: G and tail must be 16-byte aligned");
  TCE_CHECK_ARG((nl * Q * DC) % 4 == 0, "tce_mask_tail_f32: G rows must be whole 16-byte units");
  hipLaunchKernelGGL(mask_tail_kernel, dim3(tce_cdiv(h * w, 128) * tce_cdiv(nl * Q, TAIL_ITEMS), T), dim3(128), 0,
                     (hipStream_t)stream, G, tail, refs, ref_ld, masks, nl, T, Q, h, w, img_h, img_w, stride_px);
  TCE_CHECK_LAUNCH("tce_mask_tail_f32");
  return TCE_OK;
}

extern "C" int tce_copy_segments(const tceCopySeg* segs, int32_t n, tceStream stream) {
  TCE_CHECK_ARG(segs && n > 0 && n <= TCE_COPY_MAX_SEGS, "tce_copy_segments: 1..TCE_COPY_MAX_SEGS segments");
  CopySegs pack;
  long long biggest = 0;
  for (int i = 0; i < n; ++i) {
    TCE_CHECK_ARG(segs[i].src && segs[i].dst && segs[i].rows > 0 && segs[i].row_words > 0 &&
                      (segs[i].rows == 1 || segs[i].src_pitch_words >= segs[i].row_words),
                  "tce_copy_segments: bad segment");
    pack.s[i] = segs[i];
    biggest = std::max<long long>(biggest, (long long)segs[i].rows * segs[i].row_words);
  }
  const int gx = (int)std::min<long long>(1024, tce_cdiv(biggest, 1024));
  hipLaunchKernelGGL(copy_segments_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, pack);
  TCE_CHECK_LAUNCH("tce_copy_segments");
  return TCE_OK;
}

// contrastive_cal (tce_rvos.py:512-521, --contrastive): per frame, cosine similarity (nn.CosineSimilarity(dim=2, eps=1e-6)) between
// the mean of the frame's S memory rows and the clip's sentence feature.  Two launches: column sums of S / 32 row chunks per
// frame (full 1 KiB rows, one column per thread), then one workgroup per frame finishes the mean and the three dot products.
namespace {
__global__ void __launch_bounds__(256) contrastive_partial_kernel(const float* __restrict__ mem, float* __restrict__ ws, const int S,
                                                                  const int C, const int chunks) {
  const int t = blockIdx.y, p = blockIdx.x, c = threadIdx.x;
  if (c >= C) return;
  const int r0 = (int)((long long)p * S / chunks), r1 = (int)((long long)(p + 1) * S / chunks);
  const float* src = mem + ((long long)t * S + r0) * C + c;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    a0 += src[0];
    a1 += src[C];
    a2 += src[2 * (long long)C];
    a3 += src[3 * (long long)C];
    src += 4 * (long long)C;
  }
  for (; r < r1; ++r) {
    a0 += src[0];
    src += C;
  }
  ws[((long long)t * chunks + p) * C + c] = (a0 + a1) + (a2 + a3);
}
__global__ void __launch_bounds__(256) contrastive_final_kernel(const float* __restrict__ ws, const float* __restrict__ sent,
                                                                float* __restrict__ out, const int S, const int C, const int chunks,
                                                                const int frames_per_clip, const float eps) {
  __shared__ float red[3][4];
  const int t = blockIdx.x, c = threadIdx.x;
  float m = 0.f, s = 0.f;
  if (c < C) {
    for (int p = 0; p < chunks; ++p) m += ws[((long long)t * chunks + p) * C + c];
    m /= (float)S;
    s = sent[(long long)(t / frames_per_clip) * C + c];
  }
  const float ms = wave_sum(m * s), mm = wave_sum(m * m), ss = wave_sum(s * s);
  if ((c & 63) == 0) {
    red[0][c >> 6] = ms;
    red[1][c >> 6] = mm;
    red[2][c >> 6] = ss;
  }
  __syncthreads();
  if (c == 0) {
    const float dot = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float n1 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float n2 = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    // ATen cosine_similarity (PyTorch 2.x): each norm clamped separately, x.y / (max(|x|, eps) max(|y|, eps))
    out[t] = dot / (fmaxf(sqrtf(n1), eps) * fmaxf(sqrtf(n2), eps));
  }
}
}  // namespace

extern "C" int tce_contrastive_f32(const float* memory, const float* sent, float* out, float* ws, int32_t T, int32_t S, int32_t C,
                                   int32_t frames_per_clip, tceStream stream) {
  TCE_CHECK_ARG(memory && sent && out && ws && T > 0 && S > 0 && C > 0 && C <= 256 && frames_per_clip > 0 && T % frames_per_clip == 0,
                "tce_contrastive_f32: bad arguments (C <= 256, T a multiple of the clip length)");
  const int chunks = 32;
  hipLaunchKernelGGL(contrastive_partial_kernel, dim3(chunks, T), dim3(256), 0, (hipStream_t)stream, memory, ws, S, C, chunks);
  hipLaunchKernelGGL(contrastive_final_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, ws, sent, out, S, C, chunks, frames_per_clip,
                     1e-6f);
  TCE_CHECK_LAUNCH("tce_contrastive_f32");
  return TCE_OK;
}
