// Attention cores (head_dim 32, fp32).  Token counts on this path are small (49-token windows, <= 1 200
// reduced tokens, 8..40 frame tokens, 32 text tokens), the work is <1 % of the clip's FLOPs, so these are
// latency/LDS-oriented VALU kernels: one query row per lane (q and the output row live in registers), K/V
// tiles staged in LDS and read back as wave-uniform (broadcast) ds_read_b128, scores of a whole tile kept
// in registers so softmax needs one rescale per tile and no cross-lane traffic.
#include <type_traits>
#include "common.h"
#include "../../include/tce_rvos.h"
#include "../../include/tce_rvos_debug.h"

// launches KT when `flag` (the single-pass arithmetic) is set, KF otherwise: the mode is a template parameter of the kernels
#define TCE_BY_SINGLE(flag, KT, KF, ...)              \
  do {                                                \
    if (flag) hipLaunchKernelGGL(KT, __VA_ARGS__);    \
    else hipLaunchKernelGGL(KF, __VA_ARGS__);         \
  } while (0)

namespace {

constexpr int HD = 32;  // head dim

// ---------------------------------------------------------------------------------------------------
// Swin window attention.  One wavefront per (frame, window, head); lanes 0..48 own the 49 query tokens.
// The zero-padding to a multiple of 7, the cyclic shift, window partition/reverse and the final crop are
// pure index arithmetic here (no torch.roll / F.pad / window_partition copies).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) window_attn_kernel(const float* __restrict__ qkv,
                                                          const float* __restrict__ qkv_bias,
                                                          const float* __restrict__ table, float* __restrict__ out,
                                                          int T, int H, int W, int C, int nH, int shift, int nWy,
                                                          int nWx, long long total) {
  constexpr int WS = 7, NT = 49;
  __shared__ __attribute__((aligned(16))) float sK[4][NT * HD];
  __shared__ __attribute__((aligned(16))) float sV[4][NT * HD];
  __shared__ float sB[4][169];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long item = (long long)blockIdx.x * 4 + wave;  // ((t*nWy + wy)*nWx + wx)*nH + h
  const bool active = item < total;
  const int Hp = nWy * WS, Wp = nWx * WS;
  int h = 0, wx = 0, wy = 0, t = 0;
  if (active) {
    long long r = item;
    h = (int)(r % nH); r /= nH;
    wx = (int)(r % nWx); r /= nWx;
    wy = (int)(r % nWy); r /= nWy;
    t = (int)r;
  }
  const int C3 = 3 * C;
  if (active) {
    // stage K and V of the window's 49 tokens (padded tokens: the qkv bias), and this head's bias column
    for (int i = lane; i < NT * 8; i += 64) {
      const int j = i >> 3, d4 = i & 7;
      const int yy = wy * WS + j / WS, xx = wx * WS + j % WS;      // shifted-frame coordinates
      int ys = yy + shift, xs = xx + shift;                        // source (un-shifted, padded) coordinates
      if (ys >= Hp) ys -= Hp;
      if (xs >= Wp) xs -= Wp;
      f32x4 kv, vv;
      if (ys < H && xs < W) {
        const float* p = qkv + (((long long)t * H + ys) * W + xs) * C3 + h * HD + d4 * 4;
        kv = *reinterpret_cast<const f32x4*>(p + C);
        vv = *reinterpret_cast<const f32x4*>(p + 2 * C);
      } else {
        kv = *reinterpret_cast<const f32x4*>(qkv_bias + C + h * HD + d4 * 4);
        vv = *reinterpret_cast<const f32x4*>(qkv_bias + 2 * C + h * HD + d4 * 4);
      }
      *reinterpret_cast<f32x4*>(&sK[wave][j * HD + d4 * 4]) = kv;
      *reinterpret_cast<f32x4*>(&sV[wave][j * HD + d4 * 4]) = vv;
    }
    for (int i = lane; i < 169; i += 64) sB[wave][i] = table[i * nH + h];
  }
  __syncthreads();
  if (!active || lane >= NT) return;

  const int iy = lane / WS, ix = lane % WS;
  const int yy = wy * WS + iy, xx = wx * WS + ix;
  int ys = yy + shift, xs = xx + shift;
  if (ys >= Hp) ys -= Hp;
  if (xs >= Wp) xs -= Wp;
  const bool real = (ys < H && xs < W);
  const float scale = 0.17677669529663687f;  // 32^-0.5
  float q[HD];
  {
    const float* p = real ? qkv + (((long long)t * H + ys) * W + xs) * C3 + h * HD : qkv_bias + h * HD;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + d4 * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) q[d4 * 4 + j] = v[j] * scale;
    }
  }
  // region id of the -100 shift mask (built on the padded, shifted grid)
  int rid = 0;
  if (shift > 0) {
    const int ry = yy < Hp - WS ? 0 : (yy < Hp - shift ? 1 : 2);
    const int rx = xx < Wp - WS ? 0 : (xx < Wp - shift ? 1 : 2);
    rid = ry * 3 + rx;
  }
  float s[NT];
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const f32x4* kp = reinterpret_cast<const f32x4*>(&sK[wave][j * HD]);
    float a = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const f32x4 kv = kp[d4];
      a = fmaf(q[d4 * 4 + 0], kv[0], a);
      a = fmaf(q[d4 * 4 + 1], kv[1], a);
      a = fmaf(q[d4 * 4 + 2], kv[2], a);
      a = fmaf(q[d4 * 4 + 3], kv[3], a);
    }
    const int jy = j / WS, jx = j % WS;
    a += sB[wave][(iy - jy + WS - 1) * (2 * WS - 1) + (ix - jx + WS - 1)];
    if (shift > 0) {
      const int y2 = wy * WS + jy, x2 = wx * WS + jx;
      const int ry = y2 < Hp - WS ? 0 : (y2 < Hp - shift ? 1 : 2);
      const int rx = x2 < Wp - WS ? 0 : (x2 < Wp - shift ? 1 : 2);
      if (ry * 3 + rx != rid) a += -100.0f;
    }
    s[j] = a;
    mx = fmaxf(mx, a);
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    s[j] = __expf(s[j] - mx);
    l += s[j];
  }
  const float inv = 1.0f / l;
  float o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const f32x4* vp = reinterpret_cast<const f32x4*>(&sV[wave][j * HD]);
    const float pj = s[j] * inv;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const f32x4 vv = vp[d4];
      o[d4 * 4 + 0] = fmaf(pj, vv[0], o[d4 * 4 + 0]);
      o[d4 * 4 + 1] = fmaf(pj, vv[1], o[d4 * 4 + 1]);
      o[d4 * 4 + 2] = fmaf(pj, vv[2], o[d4 * 4 + 2]);
      o[d4 * 4 + 3] = fmaf(pj, vv[3], o[d4 * 4 + 3]);
    }
  }
  if (real) {
    float* po = out + (((long long)t * H + ys) * W + xs) * C + h * HD;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      f32x4 v = {o[d4 * 4], o[d4 * 4 + 1], o[d4 * 4 + 2], o[d4 * 4 + 3]};
      *reinterpret_cast<f32x4*>(po + d4 * 4) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Generic multi-head attention on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), flash-style.
//
// A wavefront owns 32 queries of one (batch, head); a workgroup = NW waves sharing the K/V tiles (32 keys) in LDS.
// Scores are computed TRANSPOSED, S^T = K Q^T (keys on the MFMA rows, queries on the lanes), so every lane holds
// 16 of the 32 key scores of ITS query in registers: the row max / sum is 15 in-register ops + one cross-half
// shuffle, and the probabilities are already the B operand of the second product O^T = V^T P^T (k-step r pairs the
// keys crow(r,0) / crow(r,1) = the two lane halves of accumulator register r) -- no LDS round trip, no transposes.
// K rows are padded to 33 floats (the A-operand read walks keys across lanes), V rows are read 32 consecutive
// floats per half-wave.  Exact fp32 arithmetic (fmaf chains), online softmax with one rescale per 32-key tile.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <int NW>
__global__ void __launch_bounds__(64 * NW) mha_mfma_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                       const float* __restrict__ V, float* __restrict__ O, int nheads,
                                                       int Lq, int Lk, int ldq, int ldk, int ldv, int ldo, long long sQ,
                                                       long long sK, long long sV, long long sO,
                                                       const uint8_t* __restrict__ kmask, float scale) {
  constexpr int KT = 32, KP = 33;
  __shared__ float sK_[KT * KP];
  __shared__ __attribute__((aligned(16))) float sV_[KT * HD];
  __shared__ float sM[KT];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int bh = blockIdx.y;
  const int b = bh / nheads, h = bh - b * nheads;
  const int q0 = (blockIdx.x * (nthr >> 6) + wave) * 32;
  const int qi = q0 + l31;
  const bool qok = qi < Lq;
  const float* Qb = Q + b * sQ + h * HD;
  const float* Kb = K + b * sK + h * HD;
  const float* Vb = V + b * sV + h * HD;
  // B operand of S^T = K Q^T: lane holds Q[q][d = 2s + lhi] * scale for s = 0..15
  float qreg[16];
  {
    float qrow[HD];
    if (qok) {
      const float* p = Qb + (long long)qi * ldq;
#pragma unroll
      for (int d4 = 0; d4 < 8; ++d4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + d4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) qrow[d4 * 4 + j] = v[j] * scale;
      }
    } else {
#pragma unroll
      for (int d = 0; d < HD; ++d) qrow[d] = 0.f;
    }
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) qreg[s2] = lhi ? qrow[2 * s2 + 1] : qrow[2 * s2];
  }
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m = -3.0e38f, l = 0.f;  // l is this lane-half's partial sum
  // K/V tiles are double-buffered through registers: the next tile's global loads are issued before the MFMAs
  // of the current one, so a key tile no longer costs a full memory round trip (with one wave per workgroup
  // nothing else on the CU hides it).
  constexpr int NLD = 256 / (64 * NW);  // float4 pairs per thread per tile
  f32x4 kreg[NLD], vreg[NLD];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = tid + u * 64 * NW;
      const int j = i >> 3, d4 = i & 7;
      const int kk = min(k0 + j, Lk - 1);  // clamped: rows past the end are masked by sM
      kreg[u] = *reinterpret_cast<const f32x4*>(Kb + (long long)kk * ldk + d4 * 4);
      vreg[u] = *reinterpret_cast<const f32x4*>(Vb + (long long)kk * ldv + d4 * 4);
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < Lk; k0 += KT) {
    const int kn = min(KT, Lk - k0);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = tid + u * 64 * NW;
      const int j = i >> 3, d4 = i & 7;
      const bool ok = j < kn;
#pragma unroll
      for (int c = 0; c < 4; ++c) sK_[j * KP + d4 * 4 + c] = ok ? kreg[u][c] : 0.f;
      f32x4 vv = vreg[u];
      if (!ok) vv = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&sV_[j * HD + d4 * 4]) = vv;
    }
    for (int j = tid; j < KT; j += nthr)
      sM[j] = (j < kn && !(kmask && kmask[(long long)b * Lk + k0 + j])) ? 0.f : -3.0e38f;
    if (k0 + KT < Lk) fetch(k0 + KT);
    __syncthreads();
    // S^T[key][q] : A = K[key = l31][d = 2s + lhi], B = qreg[s]
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(sK_[l31 * KP + 2 * s2 + lhi], qreg[s2], st, 0, 0, 0);
    float tmax = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = fminf(st[r], 3.0e38f) + sM[crow(r, lhi)];
      tmax = fmaxf(tmax, st[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(m, tmax);
    const float corr = __expf(m - mnew);
    l *= corr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o[r] *= corr;
      const float pj = (st[r] > -1.0e38f) ? __expf(st[r] - mnew) : 0.f;
      st[r] = pj;
      l += pj;
    }
    // O^T[d][q] += V^T[d][key] P^T[key][q] : k-step r pairs keys crow(r,0), crow(r,1)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(sV_[crow(r, lhi) * HD + l31], st[r], o, 0, 0, 0);
    m = mnew;
  }
  l += __shfl_xor(l, 32, 64);
  if (qok) {
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float* po = O + b * sO + (long long)qi * ldo + h * HD;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {  // registers 4g..4g+3 are 4 consecutive d: d = 8g + 4*lhi + (0..3)
      f32x4 v = {o[4 * g4] * inv, o[4 * g4 + 1] * inv, o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv};
      *reinterpret_cast<f32x4*>(po + 8 * g4 + 4 * lhi) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The same attention on the fp16 matrix cores with fp32-class accuracy ("3 x fp16 split", gemm_f16x3_kernel.h):
// the fp32 MFMA above issues at 1/16 of the fp16 rate, and the pixel decoder's self-attention over a few
// thousand tokens (segmentation.py:333-361) spends 2048 MFMA cycles per 32-key tile on it.  Here K, V (per tile,
// once per workgroup), Q (once) and the probabilities (per tile) are split into fp16 hi/lo and every product is three
// v_mfma_f32_32x32x16_f16: 2 x 6 MFMAs = 384 cycles per tile.  Same transposed formulation: S^T = K Q^T has the 32
// key scores of a query spread over the 16 accumulator registers of its two lanes, which IS the B operand of
// O^T = V^T P^T if the k order inside a 16-key step is (j&3) + 8(j>>2) + 4*half -- V^T is written to LDS so that the
// A operand of that order is two 8-byte reads.  Planes are rows of 32 halfs at an 80-byte pitch (conflict-free for the
// row-per-lane 16-byte reads).  mode 2 (tce_set_gemm_mode): one MFMA per product on nearest-rounded operands.
// ---------------------------------------------------------------------------------------------------
typedef _Float16 ah16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ah16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 afp16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void attn_split2(const float a, const float b, unsigned& hi, unsigned& lo, const int single) {
  if (single) {
    hi = __builtin_bit_cast(unsigned, afp16x2{(__fp16)a, (__fp16)b});
    lo = 0u;
    return;
  }
  const afp16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a - (float)h[0], b - (float)h[1]));
}

// SINGLE: the arithmetic mode (tce_set_gemm_mode(2) = one MFMA per product) as a COMPILE-TIME parameter of the split-fp16 attention
// kernels: as a run-time flag it put uniform branches around every hi / lo split and in front of the lo-term MFMAs, and the
// basic-block boundaries kept the scheduler from overlapping the softmax arithmetic with the MFMAs (see ffn_fused_kernel).
template <int NW, bool KSPLIT, bool SINGLE>
__global__ void __launch_bounds__(64 * NW) mha_f16x3_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                            const float* __restrict__ V, float* __restrict__ O, int nheads,
                                                            int Lq, int Lk, int ldq, int ldk, int ldv, int ldo, long long sQ,
                                                            long long sK, long long sV, long long sO,
                                                            const uint8_t* __restrict__ kmask, float scale) {
  constexpr int single = SINGLE;
  // KSPLIT: the NW waves of a workgroup own the SAME 32 queries and a quarter of the keys each (private K/V tiles, no
  // workgroup barrier inside the loop), partial (max, sum, O) merged through LDS at the end -- for long key sequences
  // with too few query tiles to fill the chip a wave's serial chain of key tiles is what bounds the launch.
  // otherwise: NW waves with 32 queries each share the K/V tiles.
  constexpr int KT = 32, PITCH = 80;  // bytes per 32-half row
  constexpr int REG = KSPLIT ? NW : 1;
  constexpr int NLOAD = KSPLIT ? 64 : 64 * NW;  // threads that load one tile
  __shared__ __attribute__((aligned(16))) unsigned char sKh_[REG][KT * PITCH], sKl_[REG][KT * PITCH], sVh_[REG][HD * PITCH],
      sVl_[REG][HD * PITCH];
  __shared__ float sM_[REG][KT];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int reg = KSPLIT ? wave : 0;
  unsigned char* const sKh = sKh_[reg];
  unsigned char* const sKl = sKl_[reg];
  unsigned char* const sVh = sVh_[reg];
  unsigned char* const sVl = sVl_[reg];
  float* const sM = sM_[reg];
  const int ltid = KSPLIT ? lane : tid;
  const int bh = blockIdx.y;
  const int b = bh / nheads, h = bh - b * nheads;
  const int q0 = KSPLIT ? blockIdx.x * 32 : (blockIdx.x * (nthr >> 6) + wave) * 32;
  const int qi = q0 + l31;
  const bool qok = qi < Lq;
  const float* Qb = Q + b * sQ + h * HD;
  const float* Kb = K + b * sK + h * HD;
  const float* Vb = V + b * sV + h * HD;
  auto tile_sync = [&]() {
    if (KSPLIT) __builtin_amdgcn_wave_barrier();  // wave-private tiles: a wave's LDS operations execute in order
    else __syncthreads();
  };
  // B operand of S^T = K Q^T: lane (query, half) holds Q[q][16s + 8*half + 0..7] * scale for the two k-steps
  typedef unsigned au32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned au32x2 __attribute__((ext_vector_type(2)));
  ah16x8 qh[2], ql[2];
  {
    const float* p = Qb + (long long)min(qi, Lq - 1) * ldq;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p + 16 * s + 8 * lhi);
      const f32x4 c = *reinterpret_cast<const f32x4*>(p + 16 * s + 8 * lhi + 4);
      unsigned hw[4], lw[4];
      attn_split2(a[0] * scale, a[1] * scale, hw[0], lw[0], single);
      attn_split2(a[2] * scale, a[3] * scale, hw[1], lw[1], single);
      attn_split2(c[0] * scale, c[1] * scale, hw[2], lw[2], single);
      attn_split2(c[2] * scale, c[3] * scale, hw[3], lw[3], single);
      qh[s] = __builtin_bit_cast(ah16x8, au32x4{hw[0], hw[1], hw[2], hw[3]});
      ql[s] = __builtin_bit_cast(ah16x8, au32x4{lw[0], lw[1], lw[2], lw[3]});
    }
  }
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m = -3.0e38f, l = 0.f;  // l is this lane-half's partial sum
  // this wave's key range (whole tiles)
  const int ntiles = (Lk + KT - 1) / KT;
  const int t_begin = KSPLIT ? (int)((long long)wave * ntiles / NW) : 0;
  const int t_end = KSPLIT ? (int)((long long)(wave + 1) * ntiles / NW) : ntiles;
  constexpr int NLD = 256 / NLOAD;  // float4 pairs per loading thread per tile
  f32x4 kreg[NLD], vreg[NLD];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = ltid + u * NLOAD;
      const int j = i >> 3, d4 = i & 7;
      const int kk = min(k0 + j, Lk - 1);  // clamped: rows past the end are masked by sM
      kreg[u] = *reinterpret_cast<const f32x4*>(Kb + (long long)kk * ldk + d4 * 4);
      vreg[u] = *reinterpret_cast<const f32x4*>(Vb + (long long)kk * ldv + d4 * 4);
    }
  };
  if (t_begin < t_end) fetch(t_begin * KT);
  for (int t = t_begin; t < t_end; ++t) {
    const int k0 = t * KT;
    const int kn = min(KT, Lk - k0);
    tile_sync();
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = ltid + u * NLOAD;
      const int j = i >> 3, d4 = i & 7;
      const bool ok = j < kn;
      f32x4 kv = kreg[u], vv = vreg[u];
      if (!ok) kv = vv = f32x4{0.f, 0.f, 0.f, 0.f};
      unsigned h0, l0, h1, l1;
      attn_split2(kv[0], kv[1], h0, l0, single);
      attn_split2(kv[2], kv[3], h1, l1, single);
      *reinterpret_cast<au32x2*>(sKh + j * PITCH + d4 * 8) = au32x2{h0, h1};  // K[key j][4 d4 .. +3]
      *reinterpret_cast<au32x2*>(sKl + j * PITCH + d4 * 8) = au32x2{l0, l1};
      // V transposed: Vt[d][key j]
      attn_split2(vv[0], vv[1], h0, l0, single);
      attn_split2(vv[2], vv[3], h1, l1, single);
      unsigned short* const vh = reinterpret_cast<unsigned short*>(sVh + (d4 * 4) * PITCH) + j;
      unsigned short* const vl = reinterpret_cast<unsigned short*>(sVl + (d4 * 4) * PITCH) + j;
      vh[0] = (unsigned short)(h0 & 0xffffu);
      vh[PITCH / 2] = (unsigned short)(h0 >> 16);
      vh[2 * (PITCH / 2)] = (unsigned short)(h1 & 0xffffu);
      vh[3 * (PITCH / 2)] = (unsigned short)(h1 >> 16);
      vl[0] = (unsigned short)(l0 & 0xffffu);
      vl[PITCH / 2] = (unsigned short)(l0 >> 16);
      vl[2 * (PITCH / 2)] = (unsigned short)(l1 & 0xffffu);
      vl[3 * (PITCH / 2)] = (unsigned short)(l1 >> 16);
    }
    for (int j = ltid; j < KT; j += NLOAD)
      sM[j] = (j < kn && !(kmask && kmask[(long long)b * Lk + k0 + j])) ? 0.f : -3.0e38f;
    if (t + 1 < t_end) fetch(k0 + KT);
    tile_sync();
    // S^T[key][q]: A = K[key = l31][16s + 8*half + 0..7]
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const ah16x8 ah = *reinterpret_cast<const ah16x8*>(sKh + l31 * PITCH + (16 * s + 8 * lhi) * 2);
      const ah16x8 al = *reinterpret_cast<const ah16x8*>(sKl + l31 * PITCH + (16 * s + 8 * lhi) * 2);
      if (!single) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[s], st, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[s], st, 0, 0, 0);
      }
      st = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s], st, 0, 0, 0);
    }
    float tmax = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = fminf(st[r], 3.0e38f) + sM[crow(r, lhi)];
      tmax = fmaxf(tmax, st[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(m, tmax);
    const float corr = __expf(m - mnew);
    l *= corr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o[r] *= corr;
      const float pj = (st[r] > -1.0e38f) ? __expf(st[r] - mnew) : 0.f;
      st[r] = pj;
      l += pj;
    }
    // O^T[d][q] += V^T[d][key] P^T[key][q]; k-step s, element j <-> accumulator register 8s + j <-> key crow(8s + j, half)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned ph[4], pl[4];
#pragma unroll
      for (int q2 = 0; q2 < 4; ++q2) attn_split2(st[8 * s + 2 * q2], st[8 * s + 2 * q2 + 1], ph[q2], pl[q2], single);
      const ah16x8 bh_ = __builtin_bit_cast(ah16x8, au32x4{ph[0], ph[1], ph[2], ph[3]});
      const ah16x8 bl_ = __builtin_bit_cast(ah16x8, au32x4{pl[0], pl[1], pl[2], pl[3]});
      // keys of elements 0..3: 16s + 4*half + 0..3; of elements 4..7: 16s + 8 + 4*half + 0..3
      const int kb = (16 * s + 4 * lhi) * 2;
      const au32x2 h0 = *reinterpret_cast<const au32x2*>(sVh + l31 * PITCH + kb);
      const au32x2 h1 = *reinterpret_cast<const au32x2*>(sVh + l31 * PITCH + kb + 16);
      const au32x2 l0 = *reinterpret_cast<const au32x2*>(sVl + l31 * PITCH + kb);
      const au32x2 l1 = *reinterpret_cast<const au32x2*>(sVl + l31 * PITCH + kb + 16);
      const ah16x8 vh_ = __builtin_bit_cast(ah16x8, au32x4{h0[0], h0[1], h1[0], h1[1]});
      const ah16x8 vl_ = __builtin_bit_cast(ah16x8, au32x4{l0[0], l0[1], l1[0], l1[1]});
      if (!single) {
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh_, bl_, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl_, bh_, o, 0, 0, 0);
      }
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh_, bh_, o, 0, 0, 0);
    }
    m = mnew;
  }
  l += __shfl_xor(l, 32, 64);
  if (KSPLIT) {
    // merge the NW partial results of these 32 queries: O = sum_w O_w e^(m_w - m*), l likewise
    __shared__ float sO[NW][HD][33];
    __shared__ float sML[NW][2][32];
#pragma unroll
    for (int r = 0; r < 16; ++r) sO[wave][crow(r, lhi)][l31] = o[r];
    if (lhi == 0) {
      sML[wave][0][l31] = m;
      sML[wave][1][l31] = l;
    }
    __syncthreads();
    if (wave != 0) return;
    float mstar = -3.0e38f;
#pragma unroll
    for (int w = 0; w < NW; ++w) mstar = fmaxf(mstar, sML[w][0][l31]);
    float f[NW];
    l = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      f[w] = __expf(sML[w][0][l31] - mstar);
      l += sML[w][1][l31] * f[w];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) acc = fmaf(sO[w][crow(r, lhi)][l31], f[w], acc);
      o[r] = acc;
    }
  }
  if (qok) {
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float* po = O + b * sO + (long long)qi * ldo + h * HD;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {  // registers 4g..4g+3 are 4 consecutive d: d = 8g + 4*lhi + (0..3)
      f32x4 v = {o[4 * g4] * inv, o[4 * g4 + 1] * inv, o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv};
      *reinterpret_cast<f32x4*>(po + 8 * g4 + 4 * lhi) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Long key sequences: K and V are split / transposed ONCE into fp16 planes in global memory
//     Kh, Kl [batch*heads][Lkp][32]   (row = key, 64 bytes)        Vh, Vl [batch*heads][Lkp/32][32 channels][32 keys]
// (Lkp = Lk rounded up to 32, padding zero) and the attention waves read their MFMA A fragments of a key tile
// straight from those planes: no LDS staging, no per-tile conversion, no barrier inside the key loop -- the staged
// kernel above re-splits every K/V tile in every workgroup, which is what its time goes to once the MFMAs are cheap.
// KSPLIT as above (4 waves = 4 key ranges of the same 32 queries, merged through LDS at the end).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mha_planes_kernel(const float* __restrict__ K, const float* __restrict__ V,
                                                         unsigned char* __restrict__ ws, int nheads, int Lk, int Lkp, int ldk,
                                                         int ldv, long long sK, long long sV, long long plane, int single) {
  // one workgroup = one 32-key tile of one (batch, head); thread (j = key, d4 = 4 channels)
  __shared__ unsigned short tvh[32][36], tvl[32][36];  // [d][key] with a padded pitch
  const int tid = threadIdx.x;
  const int bh = blockIdx.y, b = bh / nheads, h = bh - b * nheads;
  const int k0 = blockIdx.x * 32;
  const int j = tid >> 3, d4 = tid & 7;
  const int key = k0 + j;
  f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
  if (key < Lk) {
    kv = *reinterpret_cast<const f32x4*>(K + b * sK + (long long)key * ldk + h * HD + d4 * 4);
    vv = *reinterpret_cast<const f32x4*>(V + b * sV + (long long)key * ldv + h * HD + d4 * 4);
  }
  typedef unsigned au32x2 __attribute__((ext_vector_type(2)));
  unsigned h0, l0, h1, l1;
  attn_split2(kv[0], kv[1], h0, l0, single);
  attn_split2(kv[2], kv[3], h1, l1, single);
  unsigned char* const Kh = ws + (long long)bh * Lkp * 64;
  unsigned char* const Kl = Kh + plane;
  *reinterpret_cast<au32x2*>(Kh + (long long)key * 64 + d4 * 8) = au32x2{h0, h1};
  *reinterpret_cast<au32x2*>(Kl + (long long)key * 64 + d4 * 8) = au32x2{l0, l1};
  attn_split2(vv[0], vv[1], h0, l0, single);
  attn_split2(vv[2], vv[3], h1, l1, single);
  tvh[4 * d4 + 0][j] = (unsigned short)(h0 & 0xffffu);
  tvh[4 * d4 + 1][j] = (unsigned short)(h0 >> 16);
  tvh[4 * d4 + 2][j] = (unsigned short)(h1 & 0xffffu);
  tvh[4 * d4 + 3][j] = (unsigned short)(h1 >> 16);
  tvl[4 * d4 + 0][j] = (unsigned short)(l0 & 0xffffu);
  tvl[4 * d4 + 1][j] = (unsigned short)(l0 >> 16);
  tvl[4 * d4 + 2][j] = (unsigned short)(l1 & 0xffffu);
  tvl[4 * d4 + 3][j] = (unsigned short)(l1 >> 16);
  __syncthreads();
  // thread (d = tid >> 3, k4 = tid & 7): 4 consecutive keys of channel d -> 8 bytes
  const int d = tid >> 3, k4 = tid & 7;
  // V^T tile by tile: [tile][channel d][32 keys], so that a key tile is 2 KiB of contiguous rows like K's
  unsigned char* const Vh = ws + 2 * plane + ((long long)bh * (Lkp / 32) + blockIdx.x) * 2048;
  unsigned char* const Vl = Vh + plane;
  const unsigned short* rh = &tvh[d][4 * k4];
  const unsigned short* rl = &tvl[d][4 * k4];
  *reinterpret_cast<au32x2*>(Vh + d * 64 + 8 * k4) =
      au32x2{(unsigned)rh[0] | ((unsigned)rh[1] << 16), (unsigned)rh[2] | ((unsigned)rh[3] << 16)};
  *reinterpret_cast<au32x2*>(Vl + d * 64 + 8 * k4) =
      au32x2{(unsigned)rl[0] | ((unsigned)rl[1] << 16), (unsigned)rl[2] | ((unsigned)rl[3] << 16)};
}

template <bool KSPLIT, bool SINGLE>
__global__ void __launch_bounds__(256) mha_presplit_kernel(const float* __restrict__ Q, const unsigned char* __restrict__ ws,
                                                           float* __restrict__ O, int nheads, int Lq, int Lk, int Lkp, int ldq,
                                                           int ldo, long long sQ, long long sO, long long plane,
                                                           const uint8_t* __restrict__ kmask, float scale) {
  constexpr int single = SINGLE;
  constexpr int KT = 32, NW = 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int bh = blockIdx.y;
  const int b = bh / nheads, h = bh - b * nheads;
  const int q0 = KSPLIT ? blockIdx.x * 32 : (blockIdx.x * NW + wave) * 32;
  const int qi = q0 + l31;
  const bool qok = qi < Lq;
  const float* Qb = Q + b * sQ + h * HD;
  const unsigned char* const Kh = ws + (long long)bh * Lkp * 64;
  const unsigned char* const Kl = Kh + plane;
  const unsigned char* const Vh = ws + 2 * plane + (long long)bh * (Lkp / 32) * 2048;
  const unsigned char* const Vl = Vh + plane;
  typedef unsigned au32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned au32x2 __attribute__((ext_vector_type(2)));
  ah16x8 qh[2], ql[2];
  {
    const float* p = Qb + (long long)min(qi, Lq - 1) * ldq;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p + 16 * s + 8 * lhi);
      const f32x4 c = *reinterpret_cast<const f32x4*>(p + 16 * s + 8 * lhi + 4);
      unsigned hw[4], lw[4];
      attn_split2(a[0] * scale, a[1] * scale, hw[0], lw[0], single);
      attn_split2(a[2] * scale, a[3] * scale, hw[1], lw[1], single);
      attn_split2(c[0] * scale, c[1] * scale, hw[2], lw[2], single);
      attn_split2(c[2] * scale, c[3] * scale, hw[3], lw[3], single);
      qh[s] = __builtin_bit_cast(ah16x8, au32x4{hw[0], hw[1], hw[2], hw[3]});
      ql[s] = __builtin_bit_cast(ah16x8, au32x4{lw[0], lw[1], lw[2], lw[3]});
    }
  }
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m = -3.0e38f, l = 0.f;
  const int ntiles = Lkp / KT;
  const int t_begin = KSPLIT ? (int)((long long)wave * ntiles / NW) : 0;
  const int t_end = KSPLIT ? (int)((long long)(wave + 1) * ntiles / NW) : ntiles;
  // fragments of one key tile: K (hi, lo) x 2 k-steps, V^T (hi, lo) x 2 k-steps x 2 runs of 4 keys
  struct Frag {
    ah16x8 kh[2], kl[2];
    au32x2 vh[2][2], vl[2][2];
  };
  auto load_tile = [&](Frag& f, const int t) {
    const long long krow = ((long long)t * KT + l31) * 64;
    const long long vrow = (long long)t * 2048 + l31 * 64;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f.kh[s] = *reinterpret_cast<const ah16x8*>(Kh + krow + (16 * s + 8 * lhi) * 2);
      f.kl[s] = *reinterpret_cast<const ah16x8*>(Kl + krow + (16 * s + 8 * lhi) * 2);
      const int kb = (16 * s + 4 * lhi) * 2;
      f.vh[s][0] = *reinterpret_cast<const au32x2*>(Vh + vrow + kb);
      f.vh[s][1] = *reinterpret_cast<const au32x2*>(Vh + vrow + kb + 16);
      f.vl[s][0] = *reinterpret_cast<const au32x2*>(Vl + vrow + kb);
      f.vl[s][1] = *reinterpret_cast<const au32x2*>(Vl + vrow + kb + 16);
    }
  };
  Frag cur, nxt;
  if (t_begin < t_end) load_tile(cur, t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    if (t + 1 < t_end) load_tile(nxt, t + 1);
    const int k0 = t * KT;
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (!single) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.kh[s], ql[s], st, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.kl[s], qh[s], st, 0, 0, 0);
      }
      st = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.kh[s], qh[s], st, 0, 0, 0);
    }
    float tmax = -3.0e38f;
    const bool edge = k0 + KT > Lk || kmask != nullptr;  // uniform
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = fminf(st[r], 3.0e38f);
      if (edge) {
        const int key = k0 + crow(r, lhi);
        if (key >= Lk || (kmask && kmask[(long long)b * Lk + key])) v = -3.0e38f;
      }
      st[r] = v;
      tmax = fmaxf(tmax, v);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(m, tmax);
    const float corr = __expf(m - mnew);
    l *= corr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o[r] *= corr;
      const float pj = (st[r] > -1.0e38f) ? __expf(st[r] - mnew) : 0.f;
      st[r] = pj;
      l += pj;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned ph[4], pl[4];
#pragma unroll
      for (int q2 = 0; q2 < 4; ++q2) attn_split2(st[8 * s + 2 * q2], st[8 * s + 2 * q2 + 1], ph[q2], pl[q2], single);
      const ah16x8 bh_ = __builtin_bit_cast(ah16x8, au32x4{ph[0], ph[1], ph[2], ph[3]});
      const ah16x8 bl_ = __builtin_bit_cast(ah16x8, au32x4{pl[0], pl[1], pl[2], pl[3]});
      const ah16x8 vh_ = __builtin_bit_cast(ah16x8, au32x4{cur.vh[s][0][0], cur.vh[s][0][1], cur.vh[s][1][0], cur.vh[s][1][1]});
      const ah16x8 vl_ = __builtin_bit_cast(ah16x8, au32x4{cur.vl[s][0][0], cur.vl[s][0][1], cur.vl[s][1][0], cur.vl[s][1][1]});
      if (!single) {
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh_, bl_, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl_, bh_, o, 0, 0, 0);
      }
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh_, bh_, o, 0, 0, 0);
    }
    m = mnew;
    cur = nxt;
  }
  l += __shfl_xor(l, 32, 64);
  if (KSPLIT) {
    __shared__ float sO[NW][HD][33];
    __shared__ float sML[NW][2][32];
#pragma unroll
    for (int r = 0; r < 16; ++r) sO[wave][crow(r, lhi)][l31] = o[r];
    if (lhi == 0) {
      sML[wave][0][l31] = m;
      sML[wave][1][l31] = l;
    }
    __syncthreads();
    if (wave != 0) return;
    float mstar = -3.0e38f;
#pragma unroll
    for (int w = 0; w < NW; ++w) mstar = fmaxf(mstar, sML[w][0][l31]);
    float f[NW];
    l = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      f[w] = __expf(sML[w][0][l31] - mstar);
      l += sML[w][1][l31] * f[w];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) acc = fmaf(sO[w][crow(r, lhi)][l31], f[w], acc);
      o[r] = acc;
    }
  }
  if (qok) {
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float* po = O + b * sO + (long long)qi * ldo + h * HD;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 v = {o[4 * g4] * inv, o[4 * g4 + 1] * inv, o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv};
      *reinterpret_cast<f32x4*>(po + 8 * g4 + 4 * lhi) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Swin window attention on the matrix cores (exact fp32 MFMA, v_mfma_f32_32x32x2_f32), same index arithmetic as
// window_attn_kernel above.  A workgroup = 4 waves = two (frame, window, head) items; the two waves of an item own
// the query tiles 0..31 and 32..48 of the window and share its K / V (49 keys, zero-padded to 64) in LDS.  As in
// mha_mfma_kernel the scores are computed transposed, S^T = K Q^T (keys on the MFMA rows, queries on the lanes): a lane
// holds 2 x 16 key scores of ITS query in registers, the relative-position bias and the -100 shift mask are added
// there, the softmax over the 49 keys is in-register (+ one cross-half shuffle), and P^T feeds O^T = V^T P^T as the B
// operand without leaving the registers.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) window_attn_mfma_kernel(const float* __restrict__ qkv,
                                                               const float* __restrict__ qkv_bias,
                                                               const float* __restrict__ table, float* __restrict__ out,
                                                               int T, int H, int W, int C, int nH, int shift, int nWy,
                                                               int nWx, long long total) {
  constexpr int WS = 7, NT = 49, KP = 33;
  __shared__ float sK[2][64 * KP];
  __shared__ __attribute__((aligned(16))) float sV[2][64 * HD];
  __shared__ float sB[2][169];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int it = wave >> 1, qt = wave & 1;  // item slot inside the workgroup, query tile
  const int l31 = lane & 31, lhi = lane >> 5;
  const long long item = (long long)blockIdx.x * 2 + it;  // ((t*nWy + wy)*nWx + wx)*nH + h
  const bool active = item < total;
  const int Hp = nWy * WS, Wp = nWx * WS;
  int h = 0, wx = 0, wy = 0, t = 0;
  if (active) {
    long long r = item;
    h = (int)(r % nH); r /= nH;
    wx = (int)(r % nWx); r /= nWx;
    wy = (int)(r % nWy); r /= nWy;
    t = (int)r;
  }
  const int C3 = 3 * C;
  // token j of the window -> source pixel (un-shifted, padded coordinates); false = padded token (qkv = bias)
  auto src_of = [&](int j, int& ys, int& xs) {
    const int yy = wy * WS + j / WS, xx = wx * WS + j % WS;
    ys = yy + shift;
    xs = xx + shift;
    if (ys >= Hp) ys -= Hp;
    if (xs >= Wp) xs -= Wp;
    return ys < H && xs < W;
  };
  if (active) {
    // the item's two waves (128 threads) stage K and V: 64 rows x 8 float4, rows >= 49 are zero
    const int t2 = tid & 127;
    for (int i = t2; i < 64 * 8; i += 128) {
      const int j = i >> 3, d4 = i & 7;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
      if (j < NT) {
        int ys, xs;
        if (src_of(j, ys, xs)) {
          const float* p = qkv + (((long long)t * H + ys) * W + xs) * C3 + h * HD + d4 * 4;
          kv = *reinterpret_cast<const f32x4*>(p + C);
          vv = *reinterpret_cast<const f32x4*>(p + 2 * C);
        } else {
          kv = *reinterpret_cast<const f32x4*>(qkv_bias + C + h * HD + d4 * 4);
          vv = *reinterpret_cast<const f32x4*>(qkv_bias + 2 * C + h * HD + d4 * 4);
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) sK[it][j * KP + d4 * 4 + c] = kv[c];
      *reinterpret_cast<f32x4*>(&sV[it][j * HD + d4 * 4]) = vv;
    }
    for (int i = t2; i < 169; i += 128) sB[it][i] = table[i * nH + h];
  }
  __syncthreads();
  if (!active) return;

  const int qi = qt * 32 + l31;          // query index inside the window (>= 49: padding lane, never stored)
  const int qc = min(qi, NT - 1);
  int qys, qxs;
  const bool real = src_of(qc, qys, qxs) && qi < NT;
  const int iy = qc / WS, ix = qc % WS;
  const float scale = 0.17677669529663687f;  // 32^-0.5
  // B operand of S^T = K Q^T: lane holds q[d = 2s + lhi] * scale
  float qreg[16];
  {
    int ys, xs;
    const bool qreal = src_of(qc, ys, xs);
    const float* p = qreal ? qkv + (((long long)t * H + ys) * W + xs) * C3 + h * HD : qkv_bias + h * HD;
    float qrow[HD];
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + d4 * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) qrow[d4 * 4 + j] = v[j] * scale;
    }
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) qreg[s2] = lhi ? qrow[2 * s2 + 1] : qrow[2 * s2];
  }
  int rid = 0;  // region id of the -100 shift mask (built on the padded, shifted grid)
  if (shift > 0) {
    const int yy = wy * WS + iy, xx = wx * WS + ix;
    const int ry = yy < Hp - WS ? 0 : (yy < Hp - shift ? 1 : 2);
    const int rx = xx < Wp - WS ? 0 : (xx < Wp - shift ? 1 : 2);
    rid = ry * 3 + rx;
  }
  f32x16 st[2];
  float mx = -3.0e38f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
      st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[it][(kt * 32 + l31) * KP + 2 * s2 + lhi], qreg[s2], st[kt], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = kt * 32 + crow(r, lhi);  // key index inside the window
      float a = -3.0e38f;
      if (j < NT) {
        const int jy = j / WS, jx = j % WS;
        a = st[kt][r] + sB[it][(iy - jy + WS - 1) * (2 * WS - 1) + (ix - jx + WS - 1)];
        if (shift > 0) {
          const int y2 = wy * WS + jy, x2 = wx * WS + jx;
          const int ry = y2 < Hp - WS ? 0 : (y2 < Hp - shift ? 1 : 2);
          const int rx = x2 < Wp - WS ? 0 : (x2 < Wp - shift ? 1 : 2);
          if (ry * 3 + rx != rid) a += -100.0f;
        }
      }
      st[kt][r] = a;
      mx = fmaxf(mx, a);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pj = st[kt][r] > -1.0e38f ? __expf(st[kt][r] - mx) : 0.f;
      st[kt][r] = pj;
      l += pj;
    }
  l += __shfl_xor(l, 32, 64);
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  // O^T[d][q] += V^T[d][key] P^T[key][q] : k-step r pairs keys crow(r,0), crow(r,1)
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[it][(kt * 32 + crow(r, lhi)) * HD + l31], st[kt][r], o, 0, 0, 0);
  if (real) {
    const float inv = 1.0f / l;
    float* po = out + (((long long)t * H + qys) * W + qxs) * C + h * HD;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {  // registers 4g..4g+3 are 4 consecutive d: d = 8g + 4*lhi + (0..3)
      f32x4 v = {o[4 * g4] * inv, o[4 * g4 + 1] * inv, o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv};
      *reinterpret_cast<f32x4*>(po + 8 * g4 + 4 * lhi) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Video-Swin 3-D window attention (video_swin_transformer.py:71-84,138-169,215-249,316-329).
// One workgroup per (window, head); the window's K/V (<= 8*7*7 = 392 tokens x 32) and this head's column of
// the relative-position table live in LDS for the whole workgroup.  Reference quirks reproduced:
//   * windows SHRINK to the feature size when it is <= the nominal window (no padding in that dim, no shift);
//   * the bias index is relative_position_index[:N,:N] of the FULL (8,7,7) window, i.e. token n of a shrunken
//     window is re-read in full-window coordinates (n/49, (n%49)/7, n%7);
//   * padded tokens (temporal padding when T is not a multiple of the depth) carry qkv = bias and are not masked;
//   * the -100 shift mask uses region ids on the padded, shifted grid; a dim with shift 0 has one region.
// ---------------------------------------------------------------------------------------------------
struct Win3D {
  int D, H, W;        // token grid (frames, rows, cols)
  int wd, wh, ww;     // effective window
  int sd, sh, sw;     // effective shift (0 if the block is not shifted)
  int Dp, Hp, Wp;     // padded grid
  int fd, fh, fw;     // nominal (full) window: 8,7,7
};

__global__ void __launch_bounds__(256) window_attn3d_kernel(const float* __restrict__ qkv,
                                                            const float* __restrict__ qkv_bias,
                                                            const float* __restrict__ table, float* __restrict__ out,
                                                            Win3D g, int C, int nH, int table_rows) {
  // static LDS, sized for the nominal 8x7x7 window (no hipFuncAttributeMaxDynamicSharedMemorySize call on the launch path)
  constexpr int NMAX = 8 * 7 * 7, TROWS = 15 * 13 * 13;
  __shared__ __attribute__((aligned(16))) float sK[NMAX * HD];  // [N][32]
  __shared__ __attribute__((aligned(16))) float sV[NMAX * HD];  // [N][32]
  __shared__ float sB[TROWS];
  __shared__ int sSrc[NMAX];  // [N] source token row or -1 (padded)
  __shared__ int sRid[NMAX];  // [N] mask region id
  const int N = g.wd * g.wh * g.ww;
  const int tid = threadIdx.x;
  const int h = blockIdx.x % nH;
  int widx = blockIdx.x / nH;
  const int nwx = g.Wp / g.ww, nwy = g.Hp / g.wh;
  const int bx = widx % nwx; widx /= nwx;
  const int by = widx % nwy;
  const int bd = widx / nwy;
  const int C3 = 3 * C;
  for (int n = tid; n < N; n += 256) {
    const int x = n % g.ww, y = (n / g.ww) % g.wh, d = n / (g.ww * g.wh);
    const int dd = bd * g.wd + d, yy = by * g.wh + y, xx = bx * g.ww + x;  // shifted-grid coordinates
    int ds = dd + g.sd, ys = yy + g.sh, xs = xx + g.sw;
    if (ds >= g.Dp) ds -= g.Dp;
    if (ys >= g.Hp) ys -= g.Hp;
    if (xs >= g.Wp) xs -= g.Wp;
    sSrc[n] = (ds < g.D && ys < g.H && xs < g.W) ? (ds * g.H + ys) * g.W + xs : -1;
    const int rd = g.sd > 0 ? (dd < g.Dp - g.wd ? 0 : (dd < g.Dp - g.sd ? 1 : 2)) : 0;
    const int ry = g.sh > 0 ? (yy < g.Hp - g.wh ? 0 : (yy < g.Hp - g.sh ? 1 : 2)) : 0;
    const int rx = g.sw > 0 ? (xx < g.Wp - g.ww ? 0 : (xx < g.Wp - g.sw ? 1 : 2)) : 0;
    sRid[n] = (rd * 3 + ry) * 3 + rx;
  }
  for (int i = tid; i < table_rows; i += 256) sB[i] = table[(long long)i * nH + h];
  __syncthreads();
  for (int i = tid; i < N * 8; i += 256) {
    const int n = i >> 3, d4 = i & 7;
    const int srow = sSrc[n];
    f32x4 kv, vv;
    if (srow >= 0) {
      const float* p = qkv + (long long)srow * C3 + h * HD + d4 * 4;
      kv = *reinterpret_cast<const f32x4*>(p + C);
      vv = *reinterpret_cast<const f32x4*>(p + 2 * C);
    } else {
      kv = *reinterpret_cast<const f32x4*>(qkv_bias + C + h * HD + d4 * 4);
      vv = *reinterpret_cast<const f32x4*>(qkv_bias + 2 * C + h * HD + d4 * 4);
    }
    *reinterpret_cast<f32x4*>(&sK[n * HD + d4 * 4]) = kv;
    *reinterpret_cast<f32x4*>(&sV[n * HD + d4 * 4]) = vv;
  }
  __syncthreads();
  const bool masked = (g.sd | g.sh | g.sw) != 0;
  const float scale = 0.17677669529663687f;
  const int fhw = g.fh * g.fw;
  for (int q0 = 0; q0 < N; q0 += 256) {
    const int i = q0 + tid;
    if (i >= N) break;
    const int srow = sSrc[i];
    const float* qp = srow >= 0 ? qkv + (long long)srow * C3 + h * HD : qkv_bias + h * HD;
    float q[HD], o[HD];
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(qp + d4 * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) q[d4 * 4 + j] = v[j] * scale;
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    // full-window coordinates of flat index i (the [:N,:N] slicing quirk)
    const int id_ = i / fhw, iy_ = (i % fhw) / g.fw, ix_ = i % g.fw;
    const int rid = sRid[i];
    float m = -3.0e38f, l = 0.f;
    for (int k0 = 0; k0 < N; k0 += 32) {
      float s[32];
      float tmax = -3.0e38f;
#pragma unroll
      for (int jj = 0; jj < 32; ++jj) {
        const int j = k0 + jj;
        float a = -3.0e38f;
        if (j < N) {
          const f32x4* kp = reinterpret_cast<const f32x4*>(&sK[j * HD]);
          a = 0.f;
#pragma unroll
          for (int d4 = 0; d4 < 8; ++d4) {
            const f32x4 kv = kp[d4];
            a = fmaf(q[d4 * 4 + 0], kv[0], a);
            a = fmaf(q[d4 * 4 + 1], kv[1], a);
            a = fmaf(q[d4 * 4 + 2], kv[2], a);
            a = fmaf(q[d4 * 4 + 3], kv[3], a);
          }
          const int jd = j / fhw, jy = (j % fhw) / g.fw, jx = j % g.fw;
          const int ridx = ((id_ - jd + g.fd - 1) * (2 * g.fh - 1) + (iy_ - jy + g.fh - 1)) * (2 * g.fw - 1) +
                           (ix_ - jx + g.fw - 1);
          a += sB[ridx];
          if (masked && sRid[j] != rid) a += -100.0f;
        }
        s[jj] = a;
        tmax = fmaxf(tmax, a);
      }
      const float mnew = fmaxf(m, tmax);
      const float corr = __expf(m - mnew);
      l *= corr;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] *= corr;
#pragma unroll
      for (int jj = 0; jj < 32; ++jj) {
        const int j = k0 + jj;
        if (j < N) {
          const float pj = __expf(s[jj] - mnew);
          l += pj;
          const f32x4* vp = reinterpret_cast<const f32x4*>(&sV[j * HD]);
#pragma unroll
          for (int d4 = 0; d4 < 8; ++d4) {
            const f32x4 vv = vp[d4];
            o[d4 * 4 + 0] = fmaf(pj, vv[0], o[d4 * 4 + 0]);
            o[d4 * 4 + 1] = fmaf(pj, vv[1], o[d4 * 4 + 1]);
            o[d4 * 4 + 2] = fmaf(pj, vv[2], o[d4 * 4 + 2]);
            o[d4 * 4 + 3] = fmaf(pj, vv[3], o[d4 * 4 + 3]);
          }
        }
      }
      m = mnew;
    }
    if (srow >= 0) {
      const float inv = 1.0f / l;
      float* po = out + (long long)srow * C + h * HD;
#pragma unroll
      for (int d4 = 0; d4 < 8; ++d4) {
        f32x4 v = {o[d4 * 4] * inv, o[d4 * 4 + 1] * inv, o[d4 * 4 + 2] * inv, o[d4 * 4 + 3] * inv};
        *reinterpret_cast<f32x4*>(po + d4 * 4) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The same 3-D window attention on the fp16 matrix cores, fp32-class accuracy through the
// 3 x fp16 split of mha_f16x3_kernel above (mode 2: one MFMA per product).  A workgroup = one (window, head), NW waves.
// The window's K and V^T (<= 392 keys, zero-padded to a multiple of 32) are split to fp16 hi/lo planes ONCE and stay in
// LDS (K rows at an 80-byte pitch, V^T rows at NKP*2 + 8 bytes: both conflict-free for the fragment reads); the waves
// walk the query tiles (32 queries each) against all key tiles.  Scores are computed transposed (S^T = K Q^T: a lane
// owns ONE query and 16 keys of the tile in its accumulator registers), so the relative-position bias -- table index
// = code(i) - code(j) + const with code(n) = the token's full-window coordinates folded to (d*13 + y)*13 + x, the
// [:N,:N] slicing quirk included -- and the -100 region mask are added in registers, the online softmax is in-register
// + one cross-half shuffle, and P^T is the B operand of O^T = V^T P^T without leaving the registers.
// ---------------------------------------------------------------------------------------------------
// NKMAX / TROWS size the static LDS: (416, 15*13*13) = the nominal (8,7,7) video window, 134 KB, one workgroup of 8 waves per CU;
// (64, 13*13) = a 2-D Swin window (tce_window_attn_f32 runs it as a (1,7,7) window of this kernel): 20 KB, 2 waves per workgroup.
template <int NW, int NKMAX, int TROWS, bool SINGLE>
__global__ void __launch_bounds__(64 * NW) window_attn3d_mfma_kernel(const float* __restrict__ qkv,
                                                                     const float* __restrict__ qkv_bias,
                                                                     const float* __restrict__ table,
                                                                     float* __restrict__ out, Win3D g, int C, int nH,
                                                                     int table_rows, int NKP) {
  constexpr int single = SINGLE;
  typedef unsigned au32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned au32x2 __attribute__((ext_vector_type(2)));
  constexpr int KPITCH = 80, VPITCH = NKMAX * 2 + 8;
  __shared__ __attribute__((aligned(16))) unsigned char sKh[NKMAX * KPITCH], sKl[NKMAX * KPITCH], sVh[HD * VPITCH], sVl[HD * VPITCH];
  __shared__ float sB[TROWS + 1];
  __shared__ int sSrc[NKMAX];  // source token row, -1 padded token, -2 no token
  __shared__ int sCR[NKMAX];   // code | region id << 16 (natural order: the query side)
  // the key side in ACCUMULATOR order: entry kt*32 + 16*hi + r = key kt*32 + crow(r, hi), so the 16 keys of a lane's score
  // registers are 16 consecutive words (four ds_read_b128 per key tile instead of sixteen ds_read_b32)
  __shared__ __attribute__((aligned(16))) int sCodeF[NKMAX], sRidF[NKMAX];
  const int N = g.wd * g.wh * g.ww;
  const int tid = threadIdx.x, nthr = 64 * NW;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int h = blockIdx.x % nH;
  int widx = blockIdx.x / nH;
  const int nwx = g.Wp / g.ww, nwy = g.Hp / g.wh;
  const int bx = widx % nwx; widx /= nwx;
  const int by = widx % nwy;
  const int bd = widx / nwy;
  const int C3 = 3 * C;
  const int fhw = g.fh * g.fw;
  for (int n = tid; n < NKP; n += nthr) {
    int src = -2, cr = 0;
    if (n < N) {
      const int x = n % g.ww, y = (n / g.ww) % g.wh, d = n / (g.ww * g.wh);
      const int dd = bd * g.wd + d, yy = by * g.wh + y, xx = bx * g.ww + x;  // shifted-grid coordinates
      int ds = dd + g.sd, ys = yy + g.sh, xs = xx + g.sw;
      if (ds >= g.Dp) ds -= g.Dp;
      if (ys >= g.Hp) ys -= g.Hp;
      if (xs >= g.Wp) xs -= g.Wp;
      src = (ds < g.D && ys < g.H && xs < g.W) ? (ds * g.H + ys) * g.W + xs : -1;
      const int rd = g.sd > 0 ? (dd < g.Dp - g.wd ? 0 : (dd < g.Dp - g.sd ? 1 : 2)) : 0;
      const int ry = g.sh > 0 ? (yy < g.Hp - g.wh ? 0 : (yy < g.Hp - g.sh ? 1 : 2)) : 0;
      const int rx = g.sw > 0 ? (xx < g.Wp - g.ww ? 0 : (xx < g.Wp - g.sw ? 1 : 2)) : 0;
      // full-window coordinates of flat index n (relative_position_index[:N,:N] of the nominal window)
      const int code = ((n / fhw) * (2 * g.fh - 1) + (n % fhw) / g.fw) * (2 * g.fw - 1) + n % g.fw;
      cr = code | (((rd * 3 + ry) * 3 + rx) << 16);
    }
    sSrc[n] = src;
    sCR[n] = cr;
    const int w32 = n & 31, hi_ = (w32 >> 2) & 1, r_ = (w32 & 3) + 4 * (w32 >> 3);  // n = kt*32 + crow(r_, hi_)
    sCodeF[(n & ~31) + 16 * hi_ + r_] = cr & 0xffff;
    sRidF[(n & ~31) + 16 * hi_ + r_] = cr >> 16;
  }
  // the table is kept pre-multiplied by log2(e): the softmax below runs in base 2 (v_exp_f32 IS 2^x; one multiply per score saved)
  for (int i = tid; i < table_rows; i += nthr) sB[i] = table[(long long)i * nH + h] * 1.4426950408889634f;
  __syncthreads();
  // K -> [key][32 halfs] hi / lo planes, V -> transposed [d][key] hi / lo planes; rows past N are zero
  for (int i = tid; i < NKP * 8; i += nthr) {
    const int n = i >> 3, d4 = i & 7;
    const int srow = sSrc[n];
    f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
    if (srow >= 0) {
      const float* p = qkv + (long long)srow * C3 + h * HD + d4 * 4;
      kv = *reinterpret_cast<const f32x4*>(p + C);
      vv = *reinterpret_cast<const f32x4*>(p + 2 * C);
    } else if (srow == -1) {
      kv = *reinterpret_cast<const f32x4*>(qkv_bias + C + h * HD + d4 * 4);
      vv = *reinterpret_cast<const f32x4*>(qkv_bias + 2 * C + h * HD + d4 * 4);
    }
    unsigned h0, l0, h1, l1;
    attn_split2(kv[0], kv[1], h0, l0, single);
    attn_split2(kv[2], kv[3], h1, l1, single);
    *reinterpret_cast<au32x2*>(sKh + n * KPITCH + d4 * 8) = au32x2{h0, h1};
    *reinterpret_cast<au32x2*>(sKl + n * KPITCH + d4 * 8) = au32x2{l0, l1};
    attn_split2(vv[0], vv[1], h0, l0, single);
    attn_split2(vv[2], vv[3], h1, l1, single);
    unsigned short* const vh = reinterpret_cast<unsigned short*>(sVh + (d4 * 4) * VPITCH) + n;
    unsigned short* const vl = reinterpret_cast<unsigned short*>(sVl + (d4 * 4) * VPITCH) + n;
    const int vp = VPITCH / 2;
    vh[0] = (unsigned short)(h0 & 0xffffu);
    vh[vp] = (unsigned short)(h0 >> 16);
    vh[2 * vp] = (unsigned short)(h1 & 0xffffu);
    vh[3 * vp] = (unsigned short)(h1 >> 16);
    vl[0] = (unsigned short)(l0 & 0xffffu);
    vl[vp] = (unsigned short)(l0 >> 16);
    vl[2 * vp] = (unsigned short)(l1 & 0xffffu);
    vl[3 * vp] = (unsigned short)(l1 >> 16);
  }
  __syncthreads();
  // the -100 mask separates the regions of the shifted grid's LAST window along each shifted dimension; every other window
  // lies in region 0 entirely (322 windows at config 3's first stage, 36 of them on the border): no mask arithmetic there
  const bool masked = (g.sd > 0 && bd == g.Dp / g.wd - 1) || (g.sh > 0 && by == nwy - 1) || (g.sw > 0 && bx == nwx - 1);
  const float scale = 0.17677669529663687f * 1.4426950408889634f;  // 32^-0.5 * log2(e): base-2 softmax
  const int koff = ((g.fd - 1) * (2 * g.fh - 1) + (g.fh - 1)) * (2 * g.fw - 1) + (g.fw - 1);
  const int nkt = NKP / 32, nqt = (N + 31) / 32;
  for (int qt = wave; qt < nqt; qt += NW) {
    const int qi = qt * 32 + l31;
    const int qc = min(qi, N - 1);
    const int srow = sSrc[qc];
    const int crq = sCR[qc];
    // table index = code_q + koff - code_k (never negative); the -100 mask applies where the region ids differ
    const int cq = (crq & 0xffff) + koff, rq = crq >> 16;
    ah16x8 qh[2], ql[2];
    {
      const float* p = srow >= 0 ? qkv + (long long)srow * C3 + h * HD : qkv_bias + h * HD;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p + 16 * s + 8 * lhi);
        const f32x4 c = *reinterpret_cast<const f32x4*>(p + 16 * s + 8 * lhi + 4);
        unsigned hw[4], lw[4];
        attn_split2(a[0] * scale, a[1] * scale, hw[0], lw[0], single);
        attn_split2(a[2] * scale, a[3] * scale, hw[1], lw[1], single);
        attn_split2(c[0] * scale, c[1] * scale, hw[2], lw[2], single);
        attn_split2(c[2] * scale, c[3] * scale, hw[3], lw[3], single);
        qh[s] = __builtin_bit_cast(ah16x8, au32x4{hw[0], hw[1], hw[2], hw[3]});
        ql[s] = __builtin_bit_cast(ah16x8, au32x4{lw[0], lw[1], lw[2], lw[3]});
      }
    }
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m = -3.0e38f, l = 0.f;
    // one key tile; MASKED / RAGGED are compile-time so that the tile body is ONE basic block (plus the rescale): a run-time test in
    // the middle keeps the scheduler from overlapping the softmax arithmetic of a tile with the MFMAs around it
    auto key_tile = [&](const int kt, auto masked_c, auto ragged_c) {
      constexpr bool MASKED = decltype(masked_c)::value, RAGGED = decltype(ragged_c)::value;
      const int k0 = kt * 32;
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const ah16x8 ah = *reinterpret_cast<const ah16x8*>(sKh + (k0 + l31) * KPITCH + (16 * s + 8 * lhi) * 2);
        const ah16x8 al = *reinterpret_cast<const ah16x8*>(sKl + (k0 + l31) * KPITCH + (16 * s + 8 * lhi) * 2);
        if (!single) {
          st = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[s], st, 0, 0, 0);
          st = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[s], st, 0, 0, 0);
        }
        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s], st, 0, 0, 0);
      }
      float tmax = -3.0e38f;
      int kc[16];
      {
        const au32x4* cp = reinterpret_cast<const au32x4*>(sCodeF + k0 + 16 * lhi);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const au32x4 c4 = cp[q4];
#pragma unroll
          for (int c = 0; c < 4; ++c) kc[4 * q4 + c] = (int)c4[c];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] += sB[cq - kc[r]];
      if (MASKED) {
        const au32x4* rp = reinterpret_cast<const au32x4*>(sRidF + k0 + 16 * lhi);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const au32x4 r4 = rp[q4];
#pragma unroll
          for (int c = 0; c < 4; ++c) st[4 * q4 + c] += (int)r4[c] != rq ? -144.26950408889634f : 0.f;  // -100 * log2(e)
        }
      }
      if (RAGGED) {  // only the last key tile can hold keys that do not exist
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + crow(r, lhi) >= N) st[r] = -3.0e38f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, st[r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float mnew = fmaxf(m, tmax);
      if (__builtin_amdgcn_ballot_w64(mnew > m) != 0) {  // the running maximum of some query moved: rescale (else corr = 1 for all)
        const float corr = __builtin_amdgcn_exp2f(m - mnew);
        l *= corr;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= corr;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pj = __builtin_amdgcn_exp2f(st[r] - mnew);  // exp2(-3e38 - m) = 0 for the keys that do not exist
        st[r] = pj;
        l += pj;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        unsigned ph[4], pl[4];
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) attn_split2(st[8 * s + 2 * q2], st[8 * s + 2 * q2 + 1], ph[q2], pl[q2], single);
        const ah16x8 bh_ = __builtin_bit_cast(ah16x8, au32x4{ph[0], ph[1], ph[2], ph[3]});
        const ah16x8 bl_ = __builtin_bit_cast(ah16x8, au32x4{pl[0], pl[1], pl[2], pl[3]});
        const int kb = (k0 + 16 * s + 4 * lhi) * 2;
        const au32x2 h0 = *reinterpret_cast<const au32x2*>(sVh + l31 * VPITCH + kb);
        const au32x2 h1 = *reinterpret_cast<const au32x2*>(sVh + l31 * VPITCH + kb + 16);
        const au32x2 l0 = *reinterpret_cast<const au32x2*>(sVl + l31 * VPITCH + kb);
        const au32x2 l1 = *reinterpret_cast<const au32x2*>(sVl + l31 * VPITCH + kb + 16);
        const ah16x8 vh_ = __builtin_bit_cast(ah16x8, au32x4{h0[0], h0[1], h1[0], h1[1]});
        const ah16x8 vl_ = __builtin_bit_cast(ah16x8, au32x4{l0[0], l0[1], l1[0], l1[1]});
        if (!single) {
          o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh_, bl_, o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl_, bh_, o, 0, 0, 0);
        }
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh_, bh_, o, 0, 0, 0);
      }
      m = mnew;
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    const bool last_ragged = nkt * 32 > N;
    if (masked) {
      for (int kt = 0; kt + 1 < nkt; ++kt) key_tile(kt, T_{}, F_{});
      if (last_ragged) key_tile(nkt - 1, T_{}, T_{});
      else key_tile(nkt - 1, T_{}, F_{});
    } else {
      for (int kt = 0; kt + 1 < nkt; ++kt) key_tile(kt, F_{}, F_{});
      if (last_ragged) key_tile(nkt - 1, F_{}, T_{});
      else key_tile(nkt - 1, F_{}, F_{});
    }
    l += __shfl_xor(l, 32, 64);
    if (qi < N && srow >= 0) {
      const float inv = 1.0f / l;
      float* po = out + (long long)srow * C + h * HD;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v = {o[4 * g4] * inv, o[4 * g4 + 1] * inv, o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv};
        *reinterpret_cast<f32x4*>(po + 8 * g4 + 4 * lhi) = v;
      }
    }
  }
}

}  // namespace

static int g_window_attn_mfma = 1;  // tuning aid (tce_debug_window_attn_set_mfma): 0 = the VALU kernel
static int g_window_attn_2d_split = 1;  // tuning aid (tce_debug_window_attn_set_mfma(2 / 3)): 0 = 2-D windows on the exact-fp32 MFMA kernel
extern "C" int tce_debug_window_attn_set_mfma(int32_t on) {  // 0 VALU kernels, 1 default, 2 = 1 with 2-D windows on the exact-fp32 MFMA kernel
  g_window_attn_mfma = on != 0;
  g_window_attn_2d_split = on != 2;
  return TCE_OK;
}

extern "C" int tce_window_attn_f32(const float* qkv, const float* qkv_bias, const float* bias_table, float* out,
                                   int32_t T, int32_t H, int32_t W, int32_t C, int32_t nH, int32_t shift,
                                   tceStream stream) {
  TCE_CHECK_ARG(qkv && qkv_bias && bias_table && out, "tce_window_attn_f32: null pointer");
  TCE_CHECK_ARG(T > 0 && H > 0 && W > 0 && nH > 0 && C == nH * 32, "tce_window_attn_f32: need C == nH*32 (C=%d nH=%d)", C,
                nH);
  TCE_CHECK_ARG(shift == 0 || shift == 3, "tce_window_attn_f32: shift must be 0 or 3 (window 7)");
  TCE_CHECK_ARG(tce_aligned16(qkv) && tce_aligned16(qkv_bias) && tce_aligned16(out),
                "tce_window_attn_f32: pointers must be 16-byte aligned");
  const int nWy = (H + 6) / 7, nWx = (W + 6) / 7;
  const long long total = (long long)T * nWy * nWx * nH;
  if (g_window_attn_mfma && g_window_attn_2d_split && tce_get_gemm_mode() != 0) {
    // split-fp16 modes: a 2-D Swin window IS a (1,7,7) window of the 3-D kernel (same roll, same padding rule -- a padded token
    // carries qkv = bias --, same region mask; relative-position index (dy+6)*13 + (dx+6) = the kernel's code difference with the
    // nominal window (1,7,7)): 49 keys in two tiles, two waves per (window, head), 20 KB of LDS -- five times fewer MFMA cycles than
    // the exact-fp32 kernel below and no per-score index arithmetic (profiles/r04_window_attn2d.txt)
    Win3D g;
    g.D = T; g.H = H; g.W = W;
    g.wd = 1; g.wh = 7; g.ww = 7;
    g.sd = 0; g.sh = shift; g.sw = shift;
    g.Dp = T; g.Hp = nWy * 7; g.Wp = nWx * 7;
    g.fd = 1; g.fh = 7; g.fw = 7;
    TCE_CHECK_ARG(total < (1ll << 31), "tce_window_attn_f32: too many (window, head) items");
    TCE_BY_SINGLE(tce_gemm_single_pass(), (window_attn3d_mfma_kernel<2, 64, 13 * 13, true>), (window_attn3d_mfma_kernel<2, 64, 13 * 13, false>),
                  dim3((unsigned)total), dim3(128), 0, (hipStream_t)stream, qkv, qkv_bias, bias_table, out, g, C, nH, 13 * 13, 64);
    TCE_CHECK_LAUNCH("tce_window_attn_f32");
    return TCE_OK;
  }
  if (g_window_attn_mfma)
    hipLaunchKernelGGL(window_attn_mfma_kernel, dim3(tce_cdiv(total, 2)), dim3(256), 0, (hipStream_t)stream, qkv, qkv_bias,
                       bias_table, out, T, H, W, C, nH, shift, nWy, nWx, total);
  else
    hipLaunchKernelGGL(window_attn_kernel, dim3(tce_cdiv(total, 4)), dim3(256), 0, (hipStream_t)stream, qkv, qkv_bias,
                       bias_table, out, T, H, W, C, nH, shift, nWy, nWx, total);
  TCE_CHECK_LAUNCH("tce_window_attn_f32");
  return TCE_OK;
}

// A/B switch (include/tce_rvos_debug.h): 0 keeps every attention launch on the exact fp32-MFMA kernel
static int g_mha_split = 1;
extern "C" int tce_debug_mha_set_split(int32_t on) {
  g_mha_split = on;
  return TCE_OK;
}

extern "C" int tce_mha_f32(const float* Q, const float* K, const float* V, float* O, int32_t batch, int32_t nheads,
                           int32_t Lq, int32_t Lk, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, int64_t sQ,
                           int64_t sK, int64_t sV, int64_t sO, const uint8_t* kmask, float scale, tceStream stream) {
  TCE_CHECK_ARG(Q && K && V && O, "tce_mha_f32: null pointer");
  TCE_CHECK_ARG(batch > 0 && nheads > 0 && Lq > 0 && Lk > 0, "tce_mha_f32: bad sizes");
  TCE_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && sQ % 4 == 0 && sK % 4 == 0 &&
                    sV % 4 == 0 && sO % 4 == 0,
                "tce_mha_f32: leading dims / strides must be multiples of 4");
  TCE_CHECK_ARG(tce_aligned16(Q) && tce_aligned16(K) && tce_aligned16(V) && tce_aligned16(O),
                "tce_mha_f32: pointers must be 16-byte aligned");
  // 32 queries per wavefront; 4 waves share the K/V tiles when there are enough query tiles to fill the chip
  const long long tiles = (long long)tce_cdiv(Lq, 32) * batch * nheads;
  const int nw = (tiles >= 4096) ? 4 : 1;
  dim3 grid(tce_cdiv(Lq, 32 * nw), batch * nheads);
  // split-fp16 / fp16 modes: long key sequences on the fp16 matrix cores (5x fewer MFMA cycles per key tile); short ones
  // (text keys, frame tokens) are launch-bound either way and stay on the exact kernel, as does mode 0
  if (g_mha_split && tce_get_gemm_mode() != 0 && Lk >= 256) {
    const int single = tce_gemm_single_pass();
    // few query tiles against many keys: split the keys over the 4 waves of a workgroup (4x shorter serial chains)
    const bool ksplit = tiles < 4096 && Lk >= 1024;
    if (ksplit) {
      dim3 gk(tce_cdiv(Lq, 32), batch * nheads);
      TCE_BY_SINGLE(single, (mha_f16x3_kernel<4, true, true>), (mha_f16x3_kernel<4, true, false>), gk, dim3(256), 0, (hipStream_t)stream, Q,
                    K, V, O, nheads, Lq, Lk, ldq, ldk, ldv, ldo, (long long)sQ, (long long)sK, (long long)sV, (long long)sO, kmask, scale);
    } else if (nw == 4) {
      TCE_BY_SINGLE(single, (mha_f16x3_kernel<4, false, true>), (mha_f16x3_kernel<4, false, false>), grid, dim3(256), 0, (hipStream_t)stream,
                    Q, K, V, O, nheads, Lq, Lk, ldq, ldk, ldv, ldo, (long long)sQ, (long long)sK, (long long)sV, (long long)sO, kmask, scale);
    } else {
      TCE_BY_SINGLE(single, (mha_f16x3_kernel<1, false, true>), (mha_f16x3_kernel<1, false, false>), grid, dim3(64), 0, (hipStream_t)stream,
                    Q, K, V, O, nheads, Lq, Lk, ldq, ldk, ldv, ldo, (long long)sQ, (long long)sK, (long long)sV, (long long)sO, kmask, scale);
    }
    TCE_CHECK_LAUNCH("tce_mha_f32");
    return TCE_OK;
  }
  if (nw == 4)
    hipLaunchKernelGGL(mha_mfma_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, Q, K, V, O, nheads, Lq, Lk, ldq, ldk,
                       ldv, ldo, (long long)sQ, (long long)sK, (long long)sV, (long long)sO, kmask, scale);
  else
    hipLaunchKernelGGL(mha_mfma_kernel<1>, grid, dim3(64), 0, (hipStream_t)stream, Q, K, V, O, nheads, Lq, Lk, ldq, ldk,
                       ldv, ldo, (long long)sQ, (long long)sK, (long long)sV, (long long)sO, kmask, scale);
  TCE_CHECK_LAUNCH("tce_mha_f32");
  return TCE_OK;
}

extern "C" int64_t tce_mha_ws_bytes(int32_t batch, int32_t nheads, int32_t Lk) {
  const long long Lkp = (Lk + 31) / 32 * 32;
  return 4 * (long long)batch * nheads * Lkp * 64;
}

extern "C" int tce_mha_ws_f32(const float* Q, const float* K, const float* V, float* O, void* ws, int32_t batch,
                              int32_t nheads, int32_t Lq, int32_t Lk, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                              int64_t sQ, int64_t sK, int64_t sV, int64_t sO, const uint8_t* kmask, float scale,
                              tceStream stream) {
  TCE_CHECK_ARG(Q && K && V && O && ws, "tce_mha_ws_f32: null pointer");
  TCE_CHECK_ARG(batch > 0 && nheads > 0 && Lq > 0 && Lk > 0, "tce_mha_ws_f32: bad sizes");
  TCE_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && sQ % 4 == 0 && sK % 4 == 0 &&
                    sV % 4 == 0 && sO % 4 == 0,
                "tce_mha_ws_f32: leading dims / strides must be multiples of 4");
  TCE_CHECK_ARG(tce_aligned16(Q) && tce_aligned16(K) && tce_aligned16(V) && tce_aligned16(O) && tce_aligned16(ws),
                "tce_mha_ws_f32: pointers must be 16-byte aligned");
  TCE_CHECK_ARG(tce_get_gemm_mode() != 0, "tce_mha_ws_f32: split-fp16 arithmetic; in exact-fp32 mode use tce_mha_f32");
  const int Lkp = (Lk + 31) / 32 * 32;
  const long long plane = (long long)batch * nheads * Lkp * 64;
  const int single = tce_gemm_single_pass();
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(mha_planes_kernel, dim3(Lkp / 32, batch * nheads), dim3(256), 0, s, K, V, (unsigned char*)ws, nheads, Lk,
                     Lkp, ldk, ldv, (long long)sK, (long long)sV, plane, single);
  const long long tiles = (long long)tce_cdiv(Lq, 32) * batch * nheads;
  if (tiles < 4096) {  // too few query tiles to fill the chip: the 4 waves of a workgroup split the keys
    TCE_BY_SINGLE(single, (mha_presplit_kernel<true, true>), (mha_presplit_kernel<true, false>), dim3(tce_cdiv(Lq, 32), batch * nheads),
                  dim3(256), 0, s, Q, (const unsigned char*)ws, O, nheads, Lq, Lk, Lkp, ldq, ldo, (long long)sQ, (long long)sO, plane, kmask,
                  scale);
  } else {
    TCE_BY_SINGLE(single, (mha_presplit_kernel<false, true>), (mha_presplit_kernel<false, false>), dim3(tce_cdiv(Lq, 128), batch * nheads),
                  dim3(256), 0, s, Q, (const unsigned char*)ws, O, nheads, Lq, Lk, Lkp, ldq, ldo, (long long)sQ, (long long)sO, plane, kmask,
                  scale);
  }
  TCE_CHECK_LAUNCH("tce_mha_ws_f32");
  return TCE_OK;
}

extern "C" int tce_window_attn3d_f32(const float* qkv, const float* qkv_bias, const float* bias_table, float* out,
                                     int32_t T, int32_t H, int32_t W, int32_t C, int32_t nH, int32_t shifted,
                                     tceStream stream) {
  TCE_CHECK_ARG(qkv && qkv_bias && bias_table && out, "tce_window_attn3d_f32: null pointer");
  TCE_CHECK_ARG(T > 0 && H > 0 && W > 0 && nH > 0 && C == nH * 32, "tce_window_attn3d_f32: need C == nH*32");
  TCE_CHECK_ARG(tce_aligned16(qkv) && tce_aligned16(qkv_bias) && tce_aligned16(out),
                "tce_window_attn3d_f32: pointers must be 16-byte aligned");
  Win3D g;
  g.D = T; g.H = H; g.W = W;
  g.fd = 8; g.fh = 7; g.fw = 7;
  const int size[3] = {T, H, W}, full[3] = {8, 7, 7};
  int win[3], sh[3], pad[3];
  for (int i = 0; i < 3; ++i) {  // get_window_size: shrink (and never shift) a dim that fits in one window
    const bool fits = size[i] <= full[i];
    win[i] = fits ? size[i] : full[i];
    sh[i] = (fits || !shifted) ? 0 : full[i] / 2;
    pad[i] = (size[i] + win[i] - 1) / win[i] * win[i];
  }
  g.wd = win[0]; g.wh = win[1]; g.ww = win[2];
  g.sd = sh[0]; g.sh = sh[1]; g.sw = sh[2];
  g.Dp = pad[0]; g.Hp = pad[1]; g.Wp = pad[2];
  const int N = g.wd * g.wh * g.ww;
  const int table_rows = (2 * 8 - 1) * 13 * 13;
  const int nwin = (g.Dp / g.wd) * (g.Hp / g.wh) * (g.Wp / g.ww);
  TCE_CHECK_ARG(N <= 392 && table_rows == 15 * 13 * 13, "tce_window_attn3d_f32: window larger than the nominal (8,7,7)");
  if (g_window_attn_mfma && tce_get_gemm_mode() != 0) {
    // fp16 matrix cores (3 x fp16 split; mode 2: single pass).  8 waves per workgroup: two per SIMD, so one wave's softmax
    // (VALU) runs under the other's MFMAs; the 13 query tiles of a full window take two rounds.
    constexpr int NW = 8;
    const int NKP = (N + 31) / 32 * 32;
    TCE_BY_SINGLE(tce_gemm_single_pass(), (window_attn3d_mfma_kernel<NW, 416, 15 * 13 * 13, true>),
                  (window_attn3d_mfma_kernel<NW, 416, 15 * 13 * 13, false>), dim3(nwin * nH), dim3(64 * NW), 0, (hipStream_t)stream, qkv,
                  qkv_bias, bias_table, out, g, C, nH, table_rows, NKP);
    TCE_CHECK_LAUNCH("tce_window_attn3d_f32");
    return TCE_OK;
  }
  hipLaunchKernelGGL(window_attn3d_kernel, dim3(nwin * nH), dim3(256), 0, (hipStream_t)stream, qkv, qkv_bias,
                     bias_table, out, g, C, nH, table_rows);
  TCE_CHECK_LAUNCH("tce_window_attn3d_f32");
  return TCE_OK;
}
