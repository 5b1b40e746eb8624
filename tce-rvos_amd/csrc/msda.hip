// Multi-scale deformable attention forward (the reference's only native op) for gfx950.
//
// Work decomposition: one half-wavefront (32 lanes) per (frame n, query q, head m); lane = channel d of the
// D = 32 wide head, so every bilinear corner fetch is one contiguous 128-byte row of value[N,S,M,D].
// The L*P (<= 16) sampling points of the (q, m) pair are distributed one per lane for the location /
// softmax arithmetic (wave shuffles for the max / sum), then broadcast lane-to-lane with ds_bpermute-free
// __shfl while all 32 lanes gather.  Value rows are served by L2 / Infinity Cache (a frame's value tensor
// is 4.9 MB at config 2).
//
// Boundary rule restated from ms_deform_im2col_cuda.cuh:34-85,438-446:
//   h_im = y*H - 0.5, w_im = x*W - 0.5; the sample counts iff h_im > -1 && w_im > -1 && h_im < H && w_im < W;
//   each of the four corners is dropped individually when it falls outside the level.
#include "common.h"
#include "../../include/tce_rvos.h"
#include "../../include/tce_rvos_debug.h"

namespace {

constexpr int D = 32;
constexpr int MAXL = 8;

// hv / wv: rows / columns of the level that are NOT padding (== H / W for an un-padded clip); vrx / vry = wv / W, hv / H:
// the reference multiplies reference points by these "valid ratios" per level and zero-fills the value rows of padded
// positions (tce_deformable_transformer.py:125-132,180; ops/modules/ms_deform_attn.py:96-97).  Rectangular (top-left)
// valid regions only: what nested_tensor_from_videos_list produces.
struct LevelInfo {
  int H[MAXL], W[MAXL], start[MAXL];
  int hv[MAXL], wv[MAXL];
  float vrx[MAXL], vry[MAXL];
};

__device__ __forceinline__ float bilinear_gather(const float* __restrict__ vbase, int Hl, int Wl, long long row_stride,
                                                 float h_im, float w_im, int Hv, int Wv) {
  // vbase points at value[n, level_start, m, d]; row_stride = M*D floats between consecutive spatial positions
  float val = 0.f;
  if (h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl) {
    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
    const float hh = 1.f - lh, hw = 1.f - lw;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    // corners outside the level contribute nothing; corners on padded positions read a zero-filled value row
    if (h_low >= 0 && w_low >= 0 && h_low < Hv && w_low < Wv) v1 = vbase[((long long)h_low * Wl + w_low) * row_stride];
    if (h_low >= 0 && w_high < Wv && h_low < Hv) v2 = vbase[((long long)h_low * Wl + w_high) * row_stride];
    if (h_high < Hv && w_low >= 0 && w_low < Wv) v3 = vbase[((long long)h_high * Wl + w_low) * row_stride];
    if (h_high < Hv && w_high < Wv) v4 = vbase[((long long)h_high * Wl + w_high) * row_stride];
    val = (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4;
  }
  return val;
}

// acc += w * (c1 v1 + c2 v2 + c3 v3 + c4 v4) for a lane's four channels, as ONE fixed chain of fused multiply-adds: every 16-byte
// form of the gather (one point at a time, several in flight, LDS-staged) gives the same bits (left to the compiler's contraction
// the forms differed in the last place)
__device__ __forceinline__ void corner_acc(f32x4& acc, const float w, const float c1, const float c2, const float c3, const float c4,
                                           const f32x4& v1, const f32x4& v2, const f32x4& v3, const f32x4& v4) {
#pragma clang fp contract(off)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float t = c1 * v1[c];
    t = __builtin_fmaf(c2, v2[c], t);
    t = __builtin_fmaf(c3, v3[c], t);
    t = __builtin_fmaf(c4, v4[c], t);
    acc[c] = __builtin_fmaf(w, t, acc[c]);
  }
}

// Fused form, dword-per-lane (fallback of msda_fused_q4_kernel for operands that are not 16-byte aligned): raw projection
// rows (offsets | logits) + reference points in, softmax and offset normalisation inside.
__global__ void __launch_bounds__(256) msda_fused_kernel(const float* __restrict__ value, const float* __restrict__ loc_or_proj,
                                                         const float* __restrict__ aw_or_ref, float* __restrict__ out,
                                                   LevelInfo lv, int N, int S, int M, int Lq, int L, int P,
                                                   int ref_dim, int ref_per_frame, long long total) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, d = lane & 31;
  // Work item = (frame n, query q, head m).  Workgroups are dealt round-robin over the 8 XCDs (ids b and b+8
  // share an L2), so with M == 8 workgroup b serves head b % 8 only: each XCD's private L2 then holds just ITS
  // head's slice of `value` (N*S*128 B = 3.1 MB at config 2, vs 24.7 MB for all heads) and the bilinear gather
  // is served by L2 instead of the Infinity Cache.  Speed only: any placement gives the same result.
  long long item;
  if (M == 8) {
    const long long pair = ((long long)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6)) * 2 + half;  // n*Lq + q
    item = pair * 8 + (blockIdx.x & 7);
    if (pair >= total / 8) item = total;
  } else {
    item = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;  // (n*Lq + q)*M + m
  }
  const bool active = item < total;
  const int LP = L * P;
  int m = 0, q = 0, n = 0;
  if (active) {
    long long r = item;
    m = (int)(r % M); r /= M;
    q = (int)(r % Lq);
    n = (int)(r / Lq);
  }
  // lane j (< LP) of the half-wave computes point j's (x, y, weight)
  const int pj = d;  // point index handled by this lane
  float px = 0.f, py = 0.f, pw = 0.f;
  if (active && pj < LP) {
    const int l = pj / P;
    {
      const int ncol = M * LP * 3;
      const float* row = loc_or_proj + ((long long)n * Lq + q) * ncol;
      const float ox = row[(m * LP + pj) * 2 + 0];
      const float oy = row[(m * LP + pj) * 2 + 1];
      pw = row[M * LP * 2 + m * LP + pj];  // logit
      const float* rp = aw_or_ref + ((long long)(ref_per_frame ? n : 0) * Lq + q) * ref_dim;
      if (ref_dim == 2) {
        px = rp[0] * lv.vrx[l] + ox / (float)lv.W[l];
        py = rp[1] * lv.vry[l] + oy / (float)lv.H[l];
      } else {
        px = rp[0] * lv.vrx[l] + ox / (float)P * (rp[2] * lv.vrx[l]) * 0.5f;
        py = rp[1] * lv.vry[l] + oy / (float)P * (rp[3] * lv.vry[l]) * 0.5f;
      }
    }
  }
  {
    // softmax over the LP logits held by lanes 0..LP-1 of each half-wave (xor-shuffles stay inside 16 lanes)
    float v = (active && pj < LP) ? pw : -3.0e38f;
    float mx = v;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float e = (active && pj < LP) ? __expf(v - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    pw = e / sum;
  }
  float acc = 0.f;
  const long long row_stride = (long long)M * D;
  const float* vn = value + ((long long)n * S) * row_stride + m * D + d;
  for (int j = 0; j < LP; ++j) {
    const int src = (half << 5) + j;
    const float x = __shfl(px, src, 64), y = __shfl(py, src, 64), w = __shfl(pw, src, 64);
    const int l = j / P;
    const int Hl = lv.H[l], Wl = lv.W[l];
    const float h_im = y * (float)Hl - 0.5f, w_im = x * (float)Wl - 0.5f;
    if (active) acc += w * bilinear_gather(vn + (long long)lv.start[l] * row_stride, Hl, Wl, row_stride, h_im, w_im, lv.hv[l], lv.wv[l]);
  }
  if (active) out[item * D + d] = acc;
}

// Fused kernel, 16-byte form: EIGHT lanes per (frame, query, head) item, a lane owns 4 of the 32 channels, so one
// wave instruction fetches eight 128-byte rows (the dword-per-lane form above is bound by the number of gather
// instructions: 6.2 M wave-instructions of 16 address-cycles each at config 2).  Lane i of an item holds sampling
// points i and i+8, which keeps the softmax reduction tree of the dword form (xor 8, 4, 2, 1) -- results are
// bit-identical.  Same head <-> XCD placement.
__global__ void __launch_bounds__(256) msda_fused_q4_kernel(const float* __restrict__ value, const float* __restrict__ proj,
                                                            const float* __restrict__ ref, float* __restrict__ out,
                                                            LevelInfo lv, int N, int S, int M, int Lq, int L, int P,
                                                            int ref_dim, int ref_per_frame, long long total) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7;                  // channel quad / point pair index inside the item
  const int gbase = lane & ~7;               // first lane of this item's 8-lane group
  long long item;
  if (M == 8) {
    const long long pair = (long long)(blockIdx.x >> 3) * 32 + (threadIdx.x >> 3);  // n*Lq + q
    item = pair * 8 + (blockIdx.x & 7);
    if (pair >= total / 8) item = total;
  } else {
    item = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);  // (n*Lq + q)*M + m
  }
  const bool active = item < total;
  const int LP = L * P;
  int m = 0, q = 0, n = 0;
  if (active) {
    long long r = item;
    m = (int)(r % M); r /= M;
    q = (int)(r % Lq);
    n = (int)(r / Lq);
  }
  float px[2] = {0.f, 0.f}, py[2] = {0.f, 0.f}, pw[2] = {-3.0e38f, -3.0e38f};
  bool have[2] = {false, false};
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int pj = sub + 8 * e;
    if (active && pj < LP) {
      have[e] = true;
      const int l = pj / P;
      const int ncol = M * LP * 3;
      const float* row = proj + ((long long)n * Lq + q) * ncol;
      const float ox = row[(m * LP + pj) * 2 + 0];
      const float oy = row[(m * LP + pj) * 2 + 1];
      pw[e] = row[M * LP * 2 + m * LP + pj];  // logit
      const float* rp = ref + ((long long)(ref_per_frame ? n : 0) * Lq + q) * ref_dim;
      if (ref_dim == 2) {
        px[e] = rp[0] * lv.vrx[l] + ox / (float)lv.W[l];
        py[e] = rp[1] * lv.vry[l] + oy / (float)lv.H[l];
      } else {
        px[e] = rp[0] * lv.vrx[l] + ox / (float)P * (rp[2] * lv.vrx[l]) * 0.5f;
        py[e] = rp[1] * lv.vry[l] + oy / (float)P * (rp[3] * lv.vry[l]) * 0.5f;
      }
    }
  }
  // softmax over the LP logits: (point i, point i+8) in-lane = the xor-8 step, then xor 4, 2, 1 across the 8 lanes
  float mx = fmaxf(pw[0], pw[1]);
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const float e0 = have[0] ? __expf(pw[0] - mx) : 0.f, e1 = have[1] ? __expf(pw[1] - mx) : 0.f;
  float sum = e0 + e1;
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float w0 = e0 / sum, w1 = e1 / sum;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long long row_stride = (long long)M * D;
  const float* vn = value + ((long long)n * S) * row_stride + m * D + sub * 4;
  for (int j = 0; j < LP; ++j) {
    const int src = gbase + (j & 7);
    const bool hi = j >= 8;  // loop-uniform
    const float x = __shfl(hi ? px[1] : px[0], src, 64), y = __shfl(hi ? py[1] : py[0], src, 64);
    const float w = __shfl(hi ? w1 : w0, src, 64);
    const int l = j / P;
    const int Hl = lv.H[l], Wl = lv.W[l];
    const float h_im = y * (float)Hl - 0.5f, w_im = x * (float)Wl - 0.5f;
    if (active && h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl) {
      const float* vbase = vn + (long long)lv.start[l] * row_stride;
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const int h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      f32x4 v1 = z, v2 = z, v3 = z, v4 = z;
      const int Hv = lv.hv[l], Wv = lv.wv[l];  // rows / columns that are not padding (their value rows read as zero)
      if (h_low >= 0 && w_low >= 0 && h_low < Hv && w_low < Wv) v1 = *reinterpret_cast<const f32x4*>(vbase + ((long long)h_low * Wl + w_low) * row_stride);
      if (h_low >= 0 && w_high < Wv && h_low < Hv) v2 = *reinterpret_cast<const f32x4*>(vbase + ((long long)h_low * Wl + w_high) * row_stride);
      if (h_high < Hv && w_low >= 0 && w_low < Wv) v3 = *reinterpret_cast<const f32x4*>(vbase + ((long long)h_high * Wl + w_low) * row_stride);
      if (h_high < Hv && w_high < Wv) v4 = *reinterpret_cast<const f32x4*>(vbase + ((long long)h_high * Wl + w_high) * row_stride);
      const float c1 = hh * hw, c2 = hh * lw, c3 = lh * hw, c4 = lh * lw;
      corner_acc(acc, w, c1, c2, c3, c4, v1, v2, v3, v4);
    }
  }
  if (active) *reinterpret_cast<f32x4*>(out + item * D + sub * 4) = acc;
}

// One sampling point's bilinear geometry, branch-free: the four corner coefficients (zero for a corner outside the level / on
// padding, all zero for a sample outside the map) and the corners' row indices inside the level, clamped so that every address is
// valid.  Same formulas as the loop forms; `0 <= c < valid extent` is ONE unsigned compare (h_high, w_high are never negative);
// 24-bit multiplies (v_mul_lo_u32 is a quarter-rate instruction).
__device__ __forceinline__ void point_geometry(const float x, const float y, const bool active, const int Hl, const int Wl,
                                               const unsigned Hv, const unsigned Wv, float (&cw)[4], unsigned (&idx)[4]) {
  const float h_im = y * (float)Hl - 0.5f, w_im = x * (float)Wl - 0.5f;
  const bool inside = active && h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl;
  const float hf = floorf(h_im), wf = floorf(w_im);
  const float lh = h_im - hf, lw = w_im - wf;
  const float hh = 1.f - lh, hw = 1.f - lw;
  // an `inside` sample has h_low in [-1, Hl - 1]; anything else takes (0, 0) for the addresses and zero coefficients
  const int h_low = inside ? (int)hf : 0, w_low = inside ? (int)wf : 0;
  const int h_high = h_low + 1, w_high = w_low + 1;
  const bool kh0 = inside && (unsigned)h_low < Hv, kh1 = inside && (unsigned)h_high < Hv;
  const bool kw0 = (unsigned)w_low < Wv, kw1 = (unsigned)w_high < Wv;
  cw[0] = (kh0 && kw0) ? hh * hw : 0.f;
  cw[1] = (kh0 && kw1) ? hh * lw : 0.f;
  cw[2] = (kh1 && kw0) ? lh * hw : 0.f;
  cw[3] = (kh1 && kw1) ? lh * lw : 0.f;
  const unsigned hl_c = (unsigned)max(h_low, 0), wl_c = (unsigned)max(w_low, 0);
  const unsigned hh_c = (unsigned)min(h_high, Hl - 1), wh_c = (unsigned)min(w_high, Wl - 1);
  const unsigned r0 = __umul24(hl_c, (unsigned)Wl), r1 = __umul24(hh_c, (unsigned)Wl);
  idx[0] = r0 + wl_c;
  idx[1] = r0 + wh_c;
  idx[2] = r1 + wl_c;
  idx[3] = r1 + wh_c;
}

// The 16-byte form with U sampling points IN FLIGHT (round 5).  The loop above walks the 16 points one after the other, every
// corner load behind a branch: a wave pays 16 memory round trips in a row, and the call runs at half of what the CUs' vector L1
// can deliver (1.54 M wave-loads of 1 KiB = 46 us at 64 B/clk per CU; measured 86 us).  Here a step computes the U points'
// corner addresses first -- branch-free: an absent corner (outside the level / on padding / a sample outside the map) keeps a
// clamped, valid address and gets the coefficient 0 -- issues the 4 U loads together and then accumulates in the SAME order with
// the SAME expression: a dropped corner contributes 0 * (a finite value) where the loop above adds c * 0, so the results are equal
// (up to the sign of a zero).  L = P = 4 (the model's shape; other shapes take the loop above).
template <int U>
__global__ void __launch_bounds__(256) msda_fused_q4u_kernel(const float* __restrict__ value, const float* __restrict__ proj,
                                                             const float* __restrict__ ref, float* __restrict__ out, LevelInfo lv,
                                                             int N, int S, int M, int Lq, int ref_dim, int ref_per_frame,
                                                             long long total) {
  constexpr int L = 4, P = 4, LP = L * P;
  static_assert(P % U == 0, "a step stays inside one level");
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7;
  const int gbase = lane & ~7;
  // (frame, query, head) of this lane's item in 32-bit arithmetic (N * Lq * M < 2^31 is checked at the launch; the 64-bit
  // divisions of the loop form above are some 150 VALU instructions per wave)
  unsigned pair, m;
  if (M == 8) {
    pair = (blockIdx.x >> 3) * 32u + (threadIdx.x >> 3);  // n*Lq + q
    m = blockIdx.x & 7u;
  } else {
    const unsigned it = blockIdx.x * 32u + (threadIdx.x >> 3);  // (n*Lq + q)*M + m
    pair = it / (unsigned)M;
    m = it - pair * (unsigned)M;
  }
  const bool active = pair < (unsigned)(total / M);
  if (!active) pair = 0, m = 0;
  const int n = (int)(pair / (unsigned)Lq), q = (int)(pair - (unsigned)n * (unsigned)Lq);
  const long long item = (long long)pair * M + m;
  float px[2] = {0.f, 0.f}, py[2] = {0.f, 0.f}, pw[2] = {-3.0e38f, -3.0e38f};
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int pj = sub + 8 * e;
    if (active) {
      const int l = pj / P;
      const int ncol = M * LP * 3;
      const float* row = proj + ((long long)n * Lq + q) * ncol;
      const float ox = row[(m * LP + pj) * 2 + 0];
      const float oy = row[(m * LP + pj) * 2 + 1];
      pw[e] = row[M * LP * 2 + m * LP + pj];  // logit
      const float* rp = ref + ((long long)(ref_per_frame ? n : 0) * Lq + q) * ref_dim;
      if (ref_dim == 2) {
        px[e] = rp[0] * lv.vrx[l] + ox / (float)lv.W[l];
        py[e] = rp[1] * lv.vry[l] + oy / (float)lv.H[l];
      } else {
        px[e] = rp[0] * lv.vrx[l] + ox / (float)P * (rp[2] * lv.vrx[l]) * 0.5f;
        py[e] = rp[1] * lv.vry[l] + oy / (float)P * (rp[3] * lv.vry[l]) * 0.5f;
      }
    }
  }
  float mx = fmaxf(pw[0], pw[1]);
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const float e0 = active ? __expf(pw[0] - mx) : 0.f, e1 = active ? __expf(pw[1] - mx) : 0.f;
  float sum = e0 + e1;
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float w0 = e0 / sum, w1 = e1 / sum;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // Addresses: the tensor's base (uniform) + a 32-bit byte offset per lane (N * S * M * D * 4 < 2^32 and M a power of two are
  // checked at the launch): no 64-bit arithmetic, no integer multiply other than two 24-bit ones per point (v_mul_lo_u32 is a
  // quarter-rate instruction: the first version of this kernel spent a quarter of its VALU cycles in 117 of them)
  const char* const vb = reinterpret_cast<const char*>(value);
  const unsigned rs_bytes_log2 = 31u - (unsigned)__builtin_clz((unsigned)(M * D * 4));
  const unsigned lane_off = (((unsigned)n * (unsigned)S) << rs_bytes_log2) + (unsigned)(m * D + sub * 4) * 4u;
  // the level loop stays rolled (unrolled, the compiler hoists all sixteen points' shuffles and address arithmetic: 118 registers
  // per lane instead of 83).  Occupancy is not what bounds the kernel: forced to 6 / 8 waves per SIMD it ran 74.7 / 82.2 us against
  // 77.1 (tools/runs/r7c.sh); the texture addressers are busy 65-72 % of the launch (TA_BUSY, tools/runs/r7f.sh) -- 1.54 M wave-loads
  // of 1 KiB at 64 B/clk per CU are 58 % of the launch on their own -- and VALU issue about as much.
#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int Hl = lv.H[l], Wl = lv.W[l];
    const unsigned Hv = (unsigned)lv.hv[l], Wv = (unsigned)lv.wv[l];
    const unsigned base_l = lane_off + ((unsigned)lv.start[l] << rs_bytes_log2);
    const bool hi = l >= 2;  // points 8..15 are the lane's second pair
    const float sx = hi ? px[1] : px[0], sy = hi ? py[1] : py[0], sw = hi ? w1 : w0;
    const int src0 = gbase + 4 * (l & 1);
#pragma unroll
    for (int p0 = 0; p0 < P; p0 += U) {
      f32x4 v[U][4];
      float cw[U][4], ww[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int src = src0 + p0 + u;
        const float x = __shfl(sx, src, 64), y = __shfl(sy, src, 64);
        ww[u] = __shfl(sw, src, 64);
        unsigned idx[4];
        point_geometry(x, y, active, Hl, Wl, Hv, Wv, cw[u], idx);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[u][k] = *reinterpret_cast<const f32x4*>(vb + (size_t)((idx[k] << rs_bytes_log2) + base_l));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) corner_acc(acc, ww[u], cw[u][0], cw[u][1], cw[u][2], cw[u][3], v[u][0], v[u][1], v[u][2], v[u][3]);
    }
  }
  if (active) *reinterpret_cast<f32x4*>(out + item * D + sub * 4) = acc;
}

// (Round 5 also built the form with the GEOMETRY computed once per point instead of once per lane -- counters, tools/runs/r7b.sh:
// these kernels are bound by VALU issue, 1955 (loop) / 1242 (U in flight) VALU instructions per wave of 4 cycles each = 88 / 56 us
// of the 86 / 78 us, and most of them are the bilinear arithmetic all 8 lanes of an item repeat.  Lane i did it for its own two
// points and the point loop broadcast nine values -- 4 coefficients, the weight, 4 row offsets -- inside the 8-lane group.
// Bit-identical, but SLOWER: 98.4 vs 77.3 us at config 2, 159 vs 122 at config 3, 348 vs 284 at config 5 -- nine ds_bpermute per
// point cost more in the CU's one LDS pipe than the ~30 VALU instructions they save on four SIMDs.  Removed.)
// LDS-staged form of the fused kernel for the encoder's self-attention (thousands of queries per frame): a workgroup owns
// one (frame, head) and a chunk of its queries, and first copies the COARSE levels of that (frame, head) value slice into
// LDS -- at config 2 levels 1..3 are 1220 rows x 128 B = 152.5 KiB of the 160 KiB -- so that three quarters of the
// 3.08 M x 4 corner reads of a call are LDS reads; level 0 (3600 rows, 450 KiB per slice) is gathered from the XCD's
// L2 as before (workgroup b serves head b % 8, so an L2 holds one head).  The staged slice is reused by every query of
// the chunk (~16x).  Arithmetic and summation order are those of msda_fused_q4_kernel: results are bit-identical.
constexpr int MSDA_LDS_BYTES = 160 * 1024;
__global__ void __launch_bounds__(1024) msda_fused_lds_kernel(const float* __restrict__ value, const float* __restrict__ proj,
                                                              const float* __restrict__ ref, float* __restrict__ out,
                                                              LevelInfo lv, int N, int S, int Lq, int L, int P, int ref_dim,
                                                              int ref_per_frame, int first_staged, int staged_rows,
                                                              int chunks, int qpc_arg) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[MSDA_LDS_BYTES];
  constexpr int M = 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int sub = lane & 7, gbase = lane & ~7;
  const int m = blockIdx.x & 7;
  const int rest = blockIdx.x >> 3;
  const int n = rest / chunks, c = rest - n * chunks;
  const int staged_row0 = lv.start[first_staged];
  // stage rows [staged_row0, staged_row0 + staged_rows): 8 threads per 128-byte row; every load of a thread is issued
  // before the first LDS store (a load -> store loop would pay one memory round trip per 16 KiB)
  {
    constexpr int NST = MSDA_LDS_BYTES / 16 / 1024;  // 10 pieces of 16 bytes per thread at most
    f32x4 st[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int i = min(tid + 1024 * k, staged_rows * 8 - 1);
      st[k] = *reinterpret_cast<const f32x4*>(value + (((long long)n * S + staged_row0 + (i >> 3)) * M + m) * D + (i & 7) * 4);
    }
#pragma unroll
    for (int k = 0; k < NST; ++k)
      if (tid + 1024 * k < staged_rows * 8) reinterpret_cast<f32x4*>(smem)[tid + 1024 * k] = st[k];
  }
  __syncthreads();
  const int qpc = qpc_arg;
  const int q0 = c * qpc, q1 = min(Lq, q0 + qpc);
  const int LP = L * P;
  const int ncol = M * LP * 3;
  for (int qb = q0; qb < q1; qb += 128) {
    const int q = qb + (tid >> 3);
    const bool active = q < q1;
    float px[2] = {0.f, 0.f}, py[2] = {0.f, 0.f}, pw[2] = {-3.0e38f, -3.0e38f};
    bool have[2] = {false, false};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int pj = sub + 8 * e;
      if (active && pj < LP) {
        have[e] = true;
        const int l = pj / P;
        const float* row = proj + ((long long)n * Lq + q) * ncol;
        const float ox = row[(m * LP + pj) * 2 + 0];
        const float oy = row[(m * LP + pj) * 2 + 1];
        pw[e] = row[M * LP * 2 + m * LP + pj];  // logit
        const float* rp = ref + ((long long)(ref_per_frame ? n : 0) * Lq + q) * ref_dim;
        if (ref_dim == 2) {
          px[e] = rp[0] * lv.vrx[l] + ox / (float)lv.W[l];
          py[e] = rp[1] * lv.vry[l] + oy / (float)lv.H[l];
        } else {
          px[e] = rp[0] * lv.vrx[l] + ox / (float)P * (rp[2] * lv.vrx[l]) * 0.5f;
          py[e] = rp[1] * lv.vry[l] + oy / (float)P * (rp[3] * lv.vry[l]) * 0.5f;
        }
      }
    }
    float mx = fmaxf(pw[0], pw[1]);
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float e0 = have[0] ? __expf(pw[0] - mx) : 0.f, e1 = have[1] ? __expf(pw[1] - mx) : 0.f;
    float sum = e0 + e1;
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float w0 = e0 / sum, w1 = e1 / sum;

    // the gather in the pipelined, branch-free form of msda_fused_q4u_kernel (L = P = 4): two points in flight, a staged level's
    // corner rows come from LDS (row index * 128 bytes + this lane's 16), the others from the XCD's L2
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 2;
    const char* const vb = reinterpret_cast<const char*>(value);
    const unsigned lane_off = (((unsigned)n * (unsigned)S) << 10) + (unsigned)(m * D + sub * 4) * 4u;  // M = 8: 1 KiB per position
    // (the level loop stays rolled: unrolled, the compiler hoists every step's loads and the 1024-thread workgroup's 128 registers
    // per lane spill -- 518 of them in the first build)
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
      const int Hl = lv.H[l], Wl = lv.W[l];
      const unsigned Hv = (unsigned)lv.hv[l], Wv = (unsigned)lv.wv[l];
      const bool hi = l >= 2;  // points 8..15 are the lane's second triple
      const float sx = hi ? px[1] : px[0], sy = hi ? py[1] : py[0], sw = hi ? w1 : w0;
      const unsigned lbase = (unsigned)(lv.start[l] - staged_row0) * 128u + (unsigned)sub * 16u;
      const unsigned gb = lane_off + ((unsigned)lv.start[l] << 10);
      const bool in_lds = l >= first_staged;  // uniform
#pragma unroll
      for (int p0 = 0; p0 < 4; p0 += U) {
        f32x4 v[U][4];
        float cw[U][4], ww[U];
        unsigned idx[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int src = gbase + 4 * (l & 1) + p0 + u;
          const float x = __shfl(sx, src, 64), y = __shfl(sy, src, 64);
          ww[u] = __shfl(sw, src, 64);
          point_geometry(x, y, active, Hl, Wl, Hv, Wv, cw[u], idx[u]);
        }
        if (in_lds) {
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = *reinterpret_cast<const f32x4*>(smem + idx[u][k] * 128u + lbase);
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = *reinterpret_cast<const f32x4*>(vb + (size_t)((idx[u][k] << 10) + gb));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) corner_acc(acc, ww[u], cw[u][0], cw[u][1], cw[u][2], cw[u][3], v[u][0], v[u][1], v[u][2], v[u][3]);
      }
    }
    if (active) *reinterpret_cast<f32x4*>(out + (((long long)n * Lq + q) * M + m) * D + sub * 4) = acc;
  }
}

// variant of the plain kernel that reads level geometry from device memory (the reference op passes
// spatial_shapes / level_start_index as device int64 tensors)
__global__ void __launch_bounds__(256) msda_plain_dev_kernel(const float* __restrict__ value,
                                                             const int64_t* __restrict__ shapes,
                                                             const int64_t* __restrict__ starts,
                                                             const float* __restrict__ loc,
                                                             const float* __restrict__ aw, float* __restrict__ out,
                                                             int N, int S, int M, int Lq, int L, int P,
                                                             long long total) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, d = lane & 31;
  const long long item = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;
  const bool active = item < total;
  const int LP = L * P;
  int m = 0, n = 0;
  if (active) {
    long long r = item;
    m = (int)(r % M); r /= M;
    n = (int)(r / Lq);
  }
  float px = 0.f, py = 0.f, pw = 0.f;
  if (active && d < LP) {
    const long long base = item * LP + d;
    px = loc[base * 2 + 0];
    py = loc[base * 2 + 1];
    pw = aw[base];
  }
  float acc = 0.f;
  const long long row_stride = (long long)M * D;
  const float* vn = value + ((long long)n * S) * row_stride + m * D + d;
  for (int j = 0; j < LP; ++j) {
    const int src = (half << 5) + j;
    const float x = __shfl(px, src, 64), y = __shfl(py, src, 64), w = __shfl(pw, src, 64);
    const int l = j / P;
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const long long st = starts[l];
    const float h_im = y * (float)Hl - 0.5f, w_im = x * (float)Wl - 0.5f;
    if (active) acc += w * bilinear_gather(vn + st * row_stride, Hl, Wl, row_stride, h_im, w_im, Hl, Wl);
  }
  if (active) out[item * D + d] = acc;
}

// 16-byte form of the plain (reference-signature) kernel: 8 lanes per (n, q, m) item, 4 channels per lane; lane i holds
// sampling points i and i+8 (L*P <= 16).  Same arithmetic per channel as msda_plain_dev_kernel.
__global__ void __launch_bounds__(256) msda_plain_q4_dev_kernel(const float* __restrict__ value,
                                                                const int64_t* __restrict__ shapes,
                                                                const int64_t* __restrict__ starts,
                                                                const float* __restrict__ loc,
                                                                const float* __restrict__ aw, float* __restrict__ out,
                                                                int N, int S, int M, int Lq, int L, int P,
                                                                long long total) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7, gbase = lane & ~7;
  const long long item = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
  const bool active = item < total;
  const int LP = L * P;
  int m = 0, n = 0;
  if (active) {
    long long r = item;
    m = (int)(r % M); r /= M;
    n = (int)(r / Lq);
  }
  float px[2] = {0.f, 0.f}, py[2] = {0.f, 0.f}, pw[2] = {0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int pj = sub + 8 * e;
    if (active && pj < LP) {
      const long long base = item * LP + pj;
      px[e] = loc[base * 2 + 0];
      py[e] = loc[base * 2 + 1];
      pw[e] = aw[base];
    }
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long long row_stride = (long long)M * D;
  const float* vn = value + ((long long)n * S) * row_stride + m * D + sub * 4;
  for (int j = 0; j < LP; ++j) {
    const int src = gbase + (j & 7);
    const bool hi = j >= 8;
    const float x = __shfl(hi ? px[1] : px[0], src, 64), y = __shfl(hi ? py[1] : py[0], src, 64);
    const float w = __shfl(hi ? pw[1] : pw[0], src, 64);
    const int l = j / P;
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const float h_im = y * (float)Hl - 0.5f, w_im = x * (float)Wl - 0.5f;
    if (active && h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl) {
      const float* vbase = vn + (long long)starts[l] * row_stride;
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const int h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      f32x4 v1 = z, v2 = z, v3 = z, v4 = z;
      if (h_low >= 0 && w_low >= 0) v1 = *reinterpret_cast<const f32x4*>(vbase + ((long long)h_low * Wl + w_low) * row_stride);
      if (h_low >= 0 && w_high <= Wl - 1) v2 = *reinterpret_cast<const f32x4*>(vbase + ((long long)h_low * Wl + w_high) * row_stride);
      if (h_high <= Hl - 1 && w_low >= 0) v3 = *reinterpret_cast<const f32x4*>(vbase + ((long long)h_high * Wl + w_low) * row_stride);
      if (h_high <= Hl - 1 && w_high <= Wl - 1) v4 = *reinterpret_cast<const f32x4*>(vbase + ((long long)h_high * Wl + w_high) * row_stride);
      const float c1 = hh * hw, c2 = hh * lw, c3 = lh * hw, c4 = lh * lw;
      corner_acc(acc, w, c1, c2, c3, c4, v1, v2, v3, v4);
    }
  }
  if (active) *reinterpret_cast<f32x4*>(out + item * D + sub * 4) = acc;
}

// Generic form of the reference op: any head dim D, any L*P.  One thread per (item, channel); the L*P sampling points
// are walked in order with the sampling locations read straight from memory (the reference kernel is templated over
// channels the same way: one thread per output scalar, ms_deform_im2col_cuda.cuh:320-455).  Used for head dims other
// than 32 (the reference's own self-test runs D = 2 and 30/64/71/..., models/ops/test.py:21-26,85-86) and L*P > 32.
__global__ void __launch_bounds__(256) msda_generic_dev_kernel(const float* __restrict__ value,
                                                               const int64_t* __restrict__ shapes,
                                                               const int64_t* __restrict__ starts,
                                                               const float* __restrict__ loc,
                                                               const float* __restrict__ aw, float* __restrict__ out,
                                                               int S, int M, int Dh, int Lq, int L, int P,
                                                               long long total_scalars) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total_scalars) return;
  const int d = (int)(idx % Dh);
  const long long item = idx / Dh;  // (n*Lq + q)*M + m
  const int m = (int)(item % M);
  const int n = (int)(item / M / Lq);
  const long long row_stride = (long long)M * Dh;
  const float* vn = value + ((long long)n * S) * row_stride + (long long)m * Dh + d;
  float acc = 0.f;
  for (int l = 0; l < L; ++l) {
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const float* vl = vn + starts[l] * row_stride;
    for (int pt = 0; pt < P; ++pt) {
      const long long base = item * (L * P) + l * P + pt;
      const float x = loc[base * 2 + 0], y = loc[base * 2 + 1], w = aw[base];
      acc += w * bilinear_gather(vl, Hl, Wl, row_stride, y * (float)Hl - 0.5f, x * (float)Wl - 0.5f, Hl, Wl);
    }
  }
  out[idx] = acc;
}

// Backward of the reference-signature op (ms_deform_attn_cuda.cu:105-186, ms_deform_im2col_cuda.cuh:87-160,457-1229):
// 32 lanes own one (n, query, head) and walk its L*P samples; a lane owns the channels c = lane, lane+32, ... .  Per
// sample and channel the reference's rule: with top = grad_out[c] * attn_weight,
//     grad_value[corner] += bilinear_weight(corner) * top                              (atomic: queries collide)
//     grad_attn          += grad_out[c] * (w1 v1 + w2 v2 + w3 v3 + w4 v4)
//     grad_loc.x         += W * top * (hh (v2 - v1) + lh (v4 - v3)),   grad_loc.y += H * top * (hw (v3 - v1) + lw (v4 - v2))
// where a corner outside the map contributes v = 0 and no atomic, and a sample outside (-1, H) x (-1, W) contributes
// nothing.  The three per-sample sums are reduced over the 32 lanes with shuffles (the reference reduces over the
// thread block in shared memory) and written by lane 0, so grad_loc / grad_attn need no zero fill.
__global__ void __launch_bounds__(256) msda_backward_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                            const int64_t* __restrict__ starts, const float* __restrict__ loc,
                                                            const float* __restrict__ aw, const float* __restrict__ gout,
                                                            float* __restrict__ gvalue, float* __restrict__ gloc,
                                                            float* __restrict__ gattn, int S, int M, int Dh, int Lq, int L, int P,
                                                            long long items) {
  const int l32 = threadIdx.x & 31;
  const long long item = ((long long)blockIdx.x * 256 + threadIdx.x) >> 5;  // (n*Lq + q)*M + m
  if (item >= items) return;  // uniform over the 32 lanes
  const int m = (int)(item % M);
  const int n = (int)(item / M / Lq);
  const long long row_stride = (long long)M * Dh;
  const long long vbase = (long long)n * S * row_stride + (long long)m * Dh;
  const float* const go = gout + item * Dh;
  for (int l = 0; l < L; ++l) {
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const long long lbase = vbase + starts[l] * row_stride;
    for (int pt = 0; pt < P; ++pt) {
      const long long sidx = item * (L * P) + l * P + pt;
      const float x = loc[2 * sidx], y = loc[2 * sidx + 1], a = aw[sidx];
      const float h_im = y * (float)Hl - 0.5f, w_im = x * (float)Wl - 0.5f;
      float g_w = 0.f, g_h = 0.f, g_a = 0.f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const bool ok1 = h_low >= 0 && w_low >= 0, ok2 = h_low >= 0 && w_high <= Wl - 1;
        const bool ok3 = h_high <= Hl - 1 && w_low >= 0, ok4 = h_high <= Hl - 1 && w_high <= Wl - 1;
        const long long p1 = lbase + ((long long)h_low * Wl + w_low) * row_stride;
        const long long p2 = p1 + row_stride, p3 = p1 + (long long)Wl * row_stride, p4 = p3 + row_stride;
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        for (int c = l32; c < Dh; c += 32) {
          const float tg = go[c], top = tg * a;
          float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
          if (ok1) {
            v1 = value[p1 + c];
            atomicAdd(gvalue + p1 + c, w1 * top);
          }
          if (ok2) {
            v2 = value[p2 + c];
            atomicAdd(gvalue + p2 + c, w2 * top);
          }
          if (ok3) {
            v3 = value[p3 + c];
            atomicAdd(gvalue + p3 + c, w3 * top);
          }
          if (ok4) {
            v4 = value[p4 + c];
            atomicAdd(gvalue + p4 + c, w4 * top);
          }
          g_a += tg * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
          g_w += top * (hh * (v2 - v1) + lh * (v4 - v3));
          g_h += top * (hw * (v3 - v1) + lw * (v4 - v2));
        }
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        g_w += __shfl_xor(g_w, o, 32);
        g_h += __shfl_xor(g_h, o, 32);
        g_a += __shfl_xor(g_a, o, 32);
      }
      if (l32 == 0) {
        gloc[2 * sidx] = (float)Wl * g_w;
        gloc[2 * sidx + 1] = (float)Hl * g_h;
        gattn[sidx] = g_a;
      }
    }
  }
}

// Few-query form (the frame tokens' and the decoder queries' calls: a few dozen (frame, query) pairs x 8 heads): one
// WAVEFRONT per (frame, query, head) item instead of 8 lanes walking the 16 sampling points one after the other -- lane =
// (point p = lane >> 2, channel octet lane & 3), so all 16 points' four corner rows (2 x 16 bytes per lane and corner) are
// in flight at once and the call costs one memory round trip instead of sixteen.  Softmax over the points and the final
// sum over them are xor-shuffles across the point index (strides 4..32).  Same bilinear rule, different summation order
// than the 8-lane form (tree over points instead of sequential).
__global__ void __launch_bounds__(256) msda_fused_fewq_kernel(const float* __restrict__ value, const float* __restrict__ proj,
                                                              const float* __restrict__ ref, float* __restrict__ out,
                                                              LevelInfo lv, int N, int S, int M, int Lq, int L, int P,
                                                              int ref_dim, int ref_per_frame, long long total) {
  const int lane = threadIdx.x & 63;
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);  // (n*Lq + q)*M + m
  if (item >= total) return;  // whole wave
  const int pj = lane >> 2, cq = lane & 3;
  const int LP = L * P;
  long long r = item;
  const int m = (int)(r % M); r /= M;
  const int q = (int)(r % Lq);
  const int n = (int)(r / Lq);
  const bool have = pj < LP;
  float px = 0.f, py = 0.f, logit = -3.0e38f;
  int l = 0;
  if (have) {
    l = pj / P;
    const float* row = proj + ((long long)n * Lq + q) * (M * LP * 3);
    const float ox = row[(m * LP + pj) * 2 + 0];
    const float oy = row[(m * LP + pj) * 2 + 1];
    logit = row[M * LP * 2 + m * LP + pj];
    const float* rp = ref + ((long long)(ref_per_frame ? n : 0) * Lq + q) * ref_dim;
    if (ref_dim == 2) {
      px = rp[0] * lv.vrx[l] + ox / (float)lv.W[l];
      py = rp[1] * lv.vry[l] + oy / (float)lv.H[l];
    } else {
      px = rp[0] * lv.vrx[l] + ox / (float)P * (rp[2] * lv.vrx[l]) * 0.5f;
      py = rp[1] * lv.vry[l] + oy / (float)P * (rp[3] * lv.vry[l]) * 0.5f;
    }
  }
  float mx = logit;
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const float e = have ? __expf(logit - mx) : 0.f;
  float sum = e;
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
  const float wgt = e / sum;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (have) {
    const int Hl = lv.H[l], Wl = lv.W[l];
    const float h_im = py * (float)Hl - 0.5f, w_im = px * (float)Wl - 0.5f;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl) {
      const long long row_stride = (long long)M * D;
      const float* vbase = value + ((long long)n * S + lv.start[l]) * row_stride + m * D + cq * 8;
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const int h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      f32x4 v[4][2] = {{z, z}, {z, z}, {z, z}, {z, z}};
      const int Hv = lv.hv[l], Wv = lv.wv[l];
      const bool ok[4] = {h_low >= 0 && w_low >= 0 && h_low < Hv && w_low < Wv, h_low >= 0 && w_high < Wv && h_low < Hv,
                          h_high < Hv && w_low >= 0 && w_low < Wv, h_high < Hv && w_high < Wv};
      const long long pos[4] = {(long long)h_low * Wl + w_low, (long long)h_low * Wl + w_high, (long long)h_high * Wl + w_low,
                                (long long)h_high * Wl + w_high};
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (ok[c]) {
          const float* pv = vbase + pos[c] * row_stride;
          v[c][0] = *reinterpret_cast<const f32x4*>(pv);
          v[c][1] = *reinterpret_cast<const f32x4*>(pv + 4);
        }
      const float c1 = hh * hw, c2 = hh * lw, c3 = lh * hw, c4 = lh * lw;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = wgt * (c1 * v[0][i >> 2][i & 3] + c2 * v[1][i >> 2][i & 3] + c3 * v[2][i >> 2][i & 3] + c4 * v[3][i >> 2][i & 3]);
    }
  }
#pragma unroll
  for (int o = 4; o < 64; o <<= 1)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
  if (pj == 0) {
    float* po = out + item * D + cq * 8;
    *reinterpret_cast<f32x4*>(po) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4*>(po + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
  }
}

// Few-query form WITHOUT a value projection of the frame ("sample, then project"; round 5).  For the frame tokens (8 per frame)
// and the decoder queries (5 per frame) the module's value = value_proj(src) is a 24100 x 256 x 256 GEMM per call at config 2
// -- eight such launches per clip, four of them on the critical path between encoder layers -- of which the call then reads
// 0.3 % of the rows.  The bilinear gather is linear in the value rows and the softmax weights of a head sum to one, so
//     out_h = sum_k a_k sum_corner cw * [corner valid] * (W_h src(corner) + b_h) = W_h s_h + b_h w_h,
//     s_h = sum_k a_k sum_corner cw [valid] src(corner)  (256 raw channels),   w_h = sum_k a_k sum_corner cw [valid]
// (a corner outside the level or on a padded position contributes neither W_h src nor b_h: the reference zero-fills the value
// rows AFTER the projection, ops/modules/ms_deform_attn.py:95-97, and grid_sample pads with zeros).  One WORKGROUP per (frame,
// query, head): wave c4 samples input channels 64 c4 .. 64 c4 + 63 of all 16 points' four corner rows (lane = point p, channel
// group of 16: every load of the call is in flight at once, as in the projected few-query form), the four waves meet in LDS and
// the workgroup applies the head's 32 x 256 slice of value_proj.weight (+ bias * w_h) in exact fp32.
__global__ void __launch_bounds__(256) msda_fewq_raw_kernel(const float* __restrict__ src, const float* __restrict__ wv,
                                                            const float* __restrict__ bv, const float* __restrict__ proj,
                                                            const float* __restrict__ ref, float* __restrict__ out, LevelInfo lv,
                                                            int N, int S, int Lq, int L, int P, int ref_dim, int ref_per_frame) {
  constexpr int M = 8, C = 256;
  __shared__ __attribute__((aligned(16))) float s_lds[C];
  __shared__ float w_lds;
  const int tid = threadIdx.x, lane = tid & 63, c4 = tid >> 6;
  const long long item = blockIdx.x;  // (n*Lq + q)*M + m
  const int pj = lane >> 2, cq = lane & 3;
  const int LP = L * P;
  long long r = item;
  const int m = (int)(r % M); r /= M;
  const int q = (int)(r % Lq);
  const int n = (int)(r / Lq);
  const bool have = pj < LP;
  float px = 0.f, py = 0.f, logit = -3.0e38f;
  int l = 0;
  if (have) {
    l = pj / P;
    const float* row = proj + ((long long)n * Lq + q) * (M * LP * 3);
    const float ox = row[(m * LP + pj) * 2 + 0];
    const float oy = row[(m * LP + pj) * 2 + 1];
    logit = row[M * LP * 2 + m * LP + pj];
    const float* rp = ref + ((long long)(ref_per_frame ? n : 0) * Lq + q) * ref_dim;
    if (ref_dim == 2) {
      px = rp[0] * lv.vrx[l] + ox / (float)lv.W[l];
      py = rp[1] * lv.vry[l] + oy / (float)lv.H[l];
    } else {
      px = rp[0] * lv.vrx[l] + ox / (float)P * (rp[2] * lv.vrx[l]) * 0.5f;
      py = rp[1] * lv.vry[l] + oy / (float)P * (rp[3] * lv.vry[l]) * 0.5f;
    }
  }
  float mx = logit;
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const float e = have ? __expf(logit - mx) : 0.f;
  float sum = e;
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
  const float wgt = e / sum;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float weff = 0.f;
  if (have) {
    const int Hl = lv.H[l], Wl = lv.W[l];
    const float h_im = py * (float)Hl - 0.5f, w_im = px * (float)Wl - 0.5f;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl) {
      const float* vbase = src + ((long long)n * S + lv.start[l]) * C + c4 * 64 + cq * 16;
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      const int h_high = h_low + 1, w_high = w_low + 1;
      const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      f32x4 v[4][4];
      const int Hv = lv.hv[l], Wv = lv.wv[l];
      const bool ok[4] = {h_low >= 0 && w_low >= 0 && h_low < Hv && w_low < Wv, h_low >= 0 && w_high < Wv && h_low < Hv,
                          h_high < Hv && w_low >= 0 && w_low < Wv, h_high < Hv && w_high < Wv};
      const long long pos[4] = {(long long)h_low * Wl + w_low, (long long)h_low * Wl + w_high, (long long)h_high * Wl + w_low,
                                (long long)h_high * Wl + w_high};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int g = 0; g < 4; ++g) v[c][g] = z;
        if (ok[c]) {
          const float* pv = vbase + pos[c] * C;
#pragma unroll
          for (int g = 0; g < 4; ++g) v[c][g] = *reinterpret_cast<const f32x4*>(pv + 4 * g);
        }
      }
      const float cw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = wgt * (cw[0] * v[0][i >> 2][i & 3] + cw[1] * v[1][i >> 2][i & 3] + cw[2] * v[2][i >> 2][i & 3] + cw[3] * v[3][i >> 2][i & 3]);
      weff = wgt * ((ok[0] ? cw[0] : 0.f) + (ok[1] ? cw[1] : 0.f) + (ok[2] ? cw[2] : 0.f) + (ok[3] ? cw[3] : 0.f));
    }
  }
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
    weff += __shfl_xor(weff, o, 64);
  }
  if (pj == 0) {
    float* ps = s_lds + c4 * 64 + cq * 16;
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(ps + 4 * g) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    if (tid == 0) w_lds = weff;  // identical in the four waves (same points, same corner tests)
  }
  __syncthreads();
  // out[32 m + d] = W[32 m + d, :] . s + b[32 m + d] * w : thread (d = tid >> 3, part = tid & 7) takes 32 input channels
  const int d = tid >> 3, part = tid & 7;
  const float* wr = wv + (long long)(32 * m + d) * C + part * 32;
  float dot = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + 4 * g);
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(s_lds + part * 32 + 4 * g);
    dot = fmaf(w4[0], s4[0], dot);
    dot = fmaf(w4[1], s4[1], dot);
    dot = fmaf(w4[2], s4[2], dot);
    dot = fmaf(w4[3], s4[3], dot);
  }
  dot += __shfl_xor(dot, 1, 64);
  dot += __shfl_xor(dot, 2, 64);
  dot += __shfl_xor(dot, 4, 64);
  if (part == 0) out[item * 32 + d] = fmaf(bv[32 * m + d], w_lds, dot);
}

}  // namespace

extern "C" int tce_ms_deform_attn_forward_f32(const float* value, const int64_t* spatial_shapes,
                                              const int64_t* level_start_index, const float* sampling_loc,
                                              const float* attn_weight, float* out, int32_t N, int32_t S, int32_t M,
                                              int32_t Dh, int32_t Lq, int32_t L, int32_t P, tceStream stream) {
  TCE_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out,
                "tce_ms_deform_attn_forward_f32: null pointer");
  TCE_CHECK_ARG(N > 0 && S > 0 && M > 0 && Dh > 0 && Lq > 0 && L > 0 && P > 0, "tce_ms_deform_attn_forward_f32: bad sizes");
  const long long total = (long long)N * Lq * M;
  if (Dh != D || L * P > 32) {  // generic head dim / point count: one thread per output scalar
    const long long scalars = total * Dh;
    hipLaunchKernelGGL(msda_generic_dev_kernel, dim3(tce_cdiv(scalars, 256)), dim3(256), 0, (hipStream_t)stream, value,
                       spatial_shapes, level_start_index, sampling_loc, attn_weight, out, S, M, Dh, Lq, L, P, scalars);
    TCE_CHECK_LAUNCH("tce_ms_deform_attn_forward_f32");
    return TCE_OK;
  }
  if (L * P <= 16 && tce_aligned16(value) && tce_aligned16(out)) {
    hipLaunchKernelGGL(msda_plain_q4_dev_kernel, dim3(tce_cdiv(total, 32)), dim3(256), 0, (hipStream_t)stream, value,
                       spatial_shapes, level_start_index, sampling_loc, attn_weight, out, N, S, M, Lq, L, P, total);
    TCE_CHECK_LAUNCH("tce_ms_deform_attn_forward_f32");
    return TCE_OK;
  }
  hipLaunchKernelGGL(msda_plain_dev_kernel, dim3(tce_cdiv(total, 8)), dim3(256), 0, (hipStream_t)stream, value,
                     spatial_shapes, level_start_index, sampling_loc, attn_weight, out, N, S, M, Lq, L, P, total);
  TCE_CHECK_LAUNCH("tce_ms_deform_attn_forward_f32");
  return TCE_OK;
}

extern "C" int tce_ms_deform_attn_backward_f32(const float* value, const int64_t* spatial_shapes,
                                               const int64_t* level_start_index, const float* sampling_loc,
                                               const float* attn_weight, const float* grad_output, float* grad_value,
                                               float* grad_sampling_loc, float* grad_attn_weight, int32_t N, int32_t S,
                                               int32_t M, int32_t Dh, int32_t Lq, int32_t L, int32_t P, tceStream stream) {
  TCE_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && grad_output && grad_value &&
                    grad_sampling_loc && grad_attn_weight,
                "tce_ms_deform_attn_backward_f32: null pointer");
  TCE_CHECK_ARG(N > 0 && S > 0 && M > 0 && Dh > 0 && Lq > 0 && L > 0 && P > 0, "tce_ms_deform_attn_backward_f32: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  // grad_value is accumulated with atomics (the reference starts from at::zeros_like(value), ms_deform_attn_cuda.cu:142)
  const hipError_t e = hipMemsetAsync(grad_value, 0, (size_t)N * S * M * Dh * sizeof(float), s);
  if (e != hipSuccess) {
    tce_set_error("tce_ms_deform_attn_backward_f32: hipMemsetAsync: %s", hipGetErrorString(e));
    return TCE_ELAUNCH;
  }
  const long long items = (long long)N * Lq * M;
  hipLaunchKernelGGL(msda_backward_kernel, dim3(tce_cdiv(items * 32, 256)), dim3(256), 0, s, value, spatial_shapes,
                     level_start_index, sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc,
                     grad_attn_weight, S, M, Dh, Lq, L, P, items);
  TCE_CHECK_LAUNCH("tce_ms_deform_attn_backward_f32");
  return TCE_OK;
}

// Measured (tools/msda_bench.py, profiles/r02_msda_lds.txt): the LDS-staged form is bit-identical and NOT faster than the
// L2-gather form (102 vs 98 us at config 2, 150 vs 151 at config 3, 396 vs 325 at config 5): the gather is bound by
// the number of wave-instructions per sample (shuffles, address arithmetic, 4 x 16-byte loads, 16 FMAs), not by where
// the rows come from.  It stays in the library, off by default (tce_debug_msda_set_lds(1) selects it; 2 = the one-point-at-a-time
// 16-byte loop, 3 = two points in flight, 0 / 4 = four (the default): A/B of round 5's forms).
static int g_msda_lds = 0;
extern "C" int tce_debug_msda_set_lds(int32_t on) {
  g_msda_lds = on;
  return TCE_OK;
}

static int g_msda_fewq = 1;
extern "C" int tce_debug_msda_set_fewq(int32_t on) {
  g_msda_fewq = on;
  return TCE_OK;
}

extern "C" int tce_msda_fused_f32(const float* value, const float* proj, const float* ref, float* out,
                                  const int32_t* shapes_hw, int32_t N, int32_t S, int32_t M, int32_t Lq, int32_t L,
                                  int32_t P, int32_t ref_dim, int32_t ref_per_frame, tceStream stream) {
  return tce_msda_fused_valid_f32(value, proj, ref, out, shapes_hw, nullptr, N, S, M, Lq, L, P, ref_dim, ref_per_frame, stream);
}

extern "C" int tce_msda_fused_valid_f32(const float* value, const float* proj, const float* ref, float* out,
                                        const int32_t* shapes_hw, const int32_t* valid_hw, int32_t N, int32_t S, int32_t M,
                                        int32_t Lq, int32_t L, int32_t P, int32_t ref_dim, int32_t ref_per_frame,
                                        tceStream stream) {
  TCE_CHECK_ARG(value && proj && ref && out && shapes_hw, "tce_msda_fused_f32: null pointer");
  TCE_CHECK_ARG(N > 0 && S > 0 && M > 0 && Lq > 0 && L > 0 && L <= MAXL && P > 0 && L * P <= 16,
                "tce_msda_fused_f32: bad sizes (L*P must be <= 16)");
  TCE_CHECK_ARG(ref_dim == 2 || ref_dim == 4, "tce_msda_fused_f32: ref_dim must be 2 or 4");
  LevelInfo lv;
  int start = 0;
  for (int l = 0; l < MAXL; ++l) {
    if (l < L) {
      lv.H[l] = shapes_hw[2 * l];
      lv.W[l] = shapes_hw[2 * l + 1];
      lv.start[l] = start;
      start += lv.H[l] * lv.W[l];
      lv.hv[l] = valid_hw ? valid_hw[2 * l] : lv.H[l];
      lv.wv[l] = valid_hw ? valid_hw[2 * l + 1] : lv.W[l];
      TCE_CHECK_ARG(lv.hv[l] >= 1 && lv.hv[l] <= lv.H[l] && lv.wv[l] >= 1 && lv.wv[l] <= lv.W[l],
                    "tce_msda_fused_f32: valid size of level %d outside 1..(H, W)", l);
      lv.vry[l] = (float)lv.hv[l] / (float)lv.H[l];  // get_valid_ratio: valid rows / rows, in fp32 like the reference
      lv.vrx[l] = (float)lv.wv[l] / (float)lv.W[l];
    } else {
      lv.H[l] = lv.W[l] = lv.hv[l] = lv.wv[l] = 1;
      lv.start[l] = 0;
      lv.vrx[l] = lv.vry[l] = 1.f;
    }
  }
  TCE_CHECK_ARG(start == S, "tce_msda_fused_f32: sum(H*W)=%d != S=%d", start, S);
  const long long total = (long long)N * Lq * M;
  if (M == 8 && Lq >= 2048 && g_msda_lds == 1 && !valid_hw && L == 4 && P == 4 && (long long)N * S * M * D < (1ll << 30) && tce_aligned16(value) && tce_aligned16(out)) {
    // LDS-staged form: the largest suffix of levels whose (frame, head) slice fits the LDS
    int first = L;
    while (first > 1 && (long long)(start - lv.start[first - 1]) * D * 4 <= MSDA_LDS_BYTES) --first;
    if (first < L) {
      const int staged_rows = start - lv.start[first];
      // queries per workgroup: a multiple of the 128 queries one pass covers, ~2 workgroups per CU over the launch
      int want = tce_cdiv(512, N * 8);
      if (want < 1) want = 1;
      const int qpc = tce_cdiv(tce_cdiv(Lq, want), 128) * 128;
      const int chunks = tce_cdiv(Lq, qpc);
      hipLaunchKernelGGL(msda_fused_lds_kernel, dim3(N * chunks * 8), dim3(1024), 0, (hipStream_t)stream, value, proj, ref,
                         out, lv, N, S, Lq, L, P, ref_dim, ref_per_frame, first, staged_rows, chunks, qpc);
      TCE_CHECK_LAUNCH("tce_msda_fused_f32(lds)");
      return TCE_OK;
    }
  }
  if (g_msda_fewq && tce_aligned16(value) && tce_aligned16(out) && total <= 8192) {  // few queries: one wave per item, all points in flight
    hipLaunchKernelGGL(msda_fused_fewq_kernel, dim3(tce_cdiv(total, 4)), dim3(256), 0, (hipStream_t)stream, value, proj, ref, out,
                       lv, N, S, M, Lq, L, P, ref_dim, ref_per_frame, total);
    TCE_CHECK_LAUNCH("tce_msda_fused_f32(fewq)");
    return TCE_OK;
  }
  if (tce_aligned16(value) && tce_aligned16(out)) {  // 16-byte form: 32 items per workgroup
    const int nb = (M == 8) ? tce_cdiv((long long)N * Lq, 32) * 8 : tce_cdiv(total, 32);
    if (L == 4 && P == 4 && g_msda_lds != 2 && (long long)N * S * M * D < (1ll << 30) && (M & (M - 1)) == 0 && S < (1 << 24) && total < (1ll << 31)) {  // U points in flight (2 = the one-by-one loop, for A/B)
      if (g_msda_lds == 3) hipLaunchKernelGGL((msda_fused_q4u_kernel<2>), dim3(nb), dim3(256), 0, (hipStream_t)stream, value, proj, ref, out, lv, N,
                                              S, M, Lq, ref_dim, ref_per_frame, total);
      else hipLaunchKernelGGL((msda_fused_q4u_kernel<4>), dim3(nb), dim3(256), 0, (hipStream_t)stream, value, proj, ref, out, lv, N, S, M, Lq,
                              ref_dim, ref_per_frame, total);
      TCE_CHECK_LAUNCH("tce_msda_fused_f32(q4u)");
      return TCE_OK;
    }
    hipLaunchKernelGGL(msda_fused_q4_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, value, proj, ref, out, lv, N, S, M,
                       Lq, L, P, ref_dim, ref_per_frame, total);
    TCE_CHECK_LAUNCH("tce_msda_fused_f32");
    return TCE_OK;
  }
  // M == 8: 8 workgroups (one per head / XCD) per group of 8 (frame, query) pairs
  const int nblocks = (M == 8) ? tce_cdiv((long long)N * Lq, 8) * 8 : tce_cdiv(total, 8);
  hipLaunchKernelGGL(msda_fused_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, value, proj, ref,
                     out, lv, N, S, M, Lq, L, P, ref_dim, ref_per_frame, total);
  TCE_CHECK_LAUNCH("tce_msda_fused_f32");
  return TCE_OK;
}

extern "C" int tce_msda_fewq_raw_f32(const float* src, const float* wv, const float* bv, const float* proj, const float* ref,
                                     float* out, const int32_t* shapes_hw, const int32_t* valid_hw, int32_t N, int32_t S, int32_t M,
                                     int32_t Lq, int32_t L, int32_t P, int32_t ref_dim, int32_t ref_per_frame, tceStream stream) {
  TCE_CHECK_ARG(src && wv && bv && proj && ref && out && shapes_hw, "tce_msda_fewq_raw_f32: null pointer");
  TCE_CHECK_ARG(N > 0 && S > 0 && M == 8 && Lq > 0 && L > 0 && L <= MAXL && P > 0 && L * P <= 16,
                "tce_msda_fewq_raw_f32: 8 heads of 32 channels (C = 256), L*P <= 16");
  TCE_CHECK_ARG(ref_dim == 2 || ref_dim == 4, "tce_msda_fewq_raw_f32: ref_dim must be 2 or 4");
  TCE_CHECK_ARG(tce_aligned16(src) && tce_aligned16(wv), "tce_msda_fewq_raw_f32: src / wv must be 16-byte aligned");
  TCE_CHECK_ARG((long long)N * Lq * M <= 65536, "tce_msda_fewq_raw_f32: the few-query form is for <= 65536 (frame, query, head) items");
  LevelInfo lv;
  int start = 0;
  for (int l = 0; l < MAXL; ++l) {
    if (l < L) {
      lv.H[l] = shapes_hw[2 * l];
      lv.W[l] = shapes_hw[2 * l + 1];
      lv.start[l] = start;
      start += lv.H[l] * lv.W[l];
      lv.hv[l] = valid_hw ? valid_hw[2 * l] : lv.H[l];
      lv.wv[l] = valid_hw ? valid_hw[2 * l + 1] : lv.W[l];
      TCE_CHECK_ARG(lv.hv[l] >= 1 && lv.hv[l] <= lv.H[l] && lv.wv[l] >= 1 && lv.wv[l] <= lv.W[l],
                    "tce_msda_fewq_raw_f32: valid size of level %d outside 1..(H, W)", l);
      lv.vry[l] = (float)lv.hv[l] / (float)lv.H[l];
      lv.vrx[l] = (float)lv.wv[l] / (float)lv.W[l];
    } else {
      lv.H[l] = lv.W[l] = lv.hv[l] = lv.wv[l] = 1;
      lv.start[l] = 0;
      lv.vrx[l] = lv.vry[l] = 1.f;
    }
  }
  TCE_CHECK_ARG(start == S, "tce_msda_fewq_raw_f32: sum(H*W)=%d != S=%d", start, S);
  hipLaunchKernelGGL(msda_fewq_raw_kernel, dim3(N * Lq * M), dim3(256), 0, (hipStream_t)stream, src, wv, bv, proj, ref, out, lv, N, S,
                     Lq, L, P, ref_dim, ref_per_frame);
  TCE_CHECK_LAUNCH("tce_msda_fewq_raw_f32");
  return TCE_OK;
}
