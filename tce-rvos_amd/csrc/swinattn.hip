// Swin attention half-block as ONE launch (VERDICT r3 "next" #4; swin_transformer.py:202-249 with WindowAttention.forward
// :127-158 inlined):
//
//     x <- x + proj( window_attention( norm1(x) ) )          for C = 96 / 128 / 192 / 256 (3 / 4 / 6 / 8 heads of 32)
//
// norm1 -> qkv projection -> S = q k^T * scale + relative-position bias (+ the -100 shift mask) -> softmax -> P v ->
// output projection -> + residual, per 7x7 window, with the zero padding to a multiple of 7 (applied AFTER norm1: padded
// tokens carry q, k, v = bias), the cyclic shift and the window partition folded into the row index arithmetic.  The
// [tokens, 3C] qkv tensor (83 MB at config 2's first stage, written and read back by the three-launch form) never exists:
// q, k, v of a window live in registers and 16 KiB of LDS.
//
// Work decomposition.  A workgroup = 4 waves = two windows; the two waves of a window own its query tokens 0..31 and
// 32..48 (a 32-token MFMA tile each; 15 lanes of the second tile idle).  All products run on the fp16 matrix cores in the
// library's arithmetic (3 x fp16 split, fp32 accumulate; one MFMA in single mode), "token on the lane":
//   q^T[d, tok]   = Wq_h  x^T      A = weight fragments, B = the wave's norm1(x) fragments        (accumulator: lane = token)
//   k^T[d, tok]   = Wk_h  x^T      the same; the accumulator IS an A operand of S^T = K Q^T (rows = keys)
//   v  [tok, d]   = x  Wv_h^T      A = the SAME x fragments, B = weight fragments (accumulator: lane = d): it IS an A
//                                  operand of O^T = V^T P^T (rows = d, k = keys) -- no transposition anywhere
//   S^T[key, tok] = K Q^T          B = q^T accumulator (k = d in accumulator-register order, the same order in both operands)
//   O^T[d, tok]   = V^T P^T        B = the softmax'ed S^T accumulators
//   out^T[c, tok] += Wp[:, h] O^T  B = O^T accumulator; Wp fragments packed in the accumulator's k order
// K and V of a head are exchanged between the two waves of a window through LDS as ready-made 1 KiB fragment pieces
// (ds_write_b128 / ds_read_b128 at lane*16: conflict-free).  The weights (static) are pre-split to fp16 hi/lo and
// pre-ordered at pack time into the exact sequence of fragments the loop consumes, per head [Wq | Wk | Wv | Wp column
// block], and stream L2 -> LDS by DMA through a two-stage ring shared by the workgroup's four waves (one barrier per stage).
// Bound: matrix cores at C >= 128 (the 49 -> 64 token padding of a window costs 23 % of the projections' MFMAs); at C = 96
// the launch is a few rounds of 4.6 us workgroups, close to the x-in / x-out HBM time (55 MB at config 2: 9 us).
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"
#include "frag.h"
#include "../../include/tce_rvos.h"

namespace {

constexpr int WS = 7, NTOK = 49;
// C up to which the kernel is built for TWO workgroups per CU (<= 256 registers, <= 80 KB of LDS: a workgroup's prologue and
// epilogue latency, and its softmax, then run under the other's MFMAs); tuning aid, see tools/swin_attn_bench.py
#ifndef SWIN_OCC2_MAXC
#define SWIN_OCC2_MAXC 128
#endif

struct SwinArgs {
  const float* x;
  float* out;
  const unsigned char* wpk;
  const float *bqkv, *bproj, *table, *g1, *be1;
  long long ldx, ldo;
  int T, H, W, nWy, nWx, shift;
  long long nwin;
  float eps;
  int* range_flag;
  int single;
  // Per key slot of a lane (slot = 16 kt + r <-> key j = 32 kt + crow(r, hf)), for hf = 0 / 1, built on the host:
  unsigned cj[2][8];    // 32 bytes: jy * 13 + jx  (relative-position index of (query, key) = (iy*13 + ix + 84) - cj)
  unsigned kvalid[2];   // bit = the key exists (j < 49)
  unsigned rowhi[2];    // bit = the key's row inside the window is >= 7 - shift (second region of the last window row)
  unsigned colhi[2];    // the same for columns
};

__device__ __forceinline__ int crow_(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// compile-time loop: f(integral_constant<int, I>) for I = 0 .. N-1 (a `#pragma unroll` over the head's 24..64 steps is not
// honoured by the optimiser; every per-step decision below must be a compile-time one: register arrays, ring offsets)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// D += A B in the library's arithmetic: three MFMAs on the hi/lo split (cross terms first), one in single mode
__device__ __forceinline__ void mma3(f32x16& d, const h16x8 ah, const h16x8 al, const h16x8 bh, const h16x8 bl, const int single) {
  if (!single) {
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, d, 0, 0, 0);
  }
  d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, d, 0, 0, 0);
}

// accumulator registers 8s..8s+7 -> fp16 hi/lo fragment of k-step s (s = 0, 1)
__device__ __forceinline__ HL acc_frag(const f32x16& a, const int s, const int single) {
  const float f[8] = {a[8 * s], a[8 * s + 1], a[8 * s + 2], a[8 * s + 3], a[8 * s + 4], a[8 * s + 5], a[8 * s + 6], a[8 * s + 7]};
  return split8(f, single);
}

template <int C>
struct SwinCfg {
  static constexpr int KS = C / 16, NTL = C / 32, NH = C / 32;
  static constexpr int HSTEPS = C / 4;             // (hi, lo) fragment pairs per head: 3 KS + 2 NTL
  static constexpr int G = (C <= SWIN_OCC2_MAXC) ? 8 : 16;    // fragment pairs per ring stage (8: 67 KB of LDS, two workgroups per CU)
  static constexpr int WG_PER_CU = (C <= SWIN_OCC2_MAXC) ? 2 : 1;
  static constexpr int IPH = HSTEPS / G;           // ring stages per head
  static constexpr int STAGE = G * 2 * PIECE;
  static constexpr int SLOTS = G * 2 / 4;          // DMAs per wave per stage
  static constexpr int EX_OFF = 2 * STAGE;         // K / V exchange: [window slot][K | V][key tile][hi0, lo0, hi1, lo1]
  static constexpr int EX_WIN = 2 * 2 * 4 * PIECE;
  static constexpr int WT_OFF = EX_OFF;            // the per-wave staging tiles (prologue / epilogue only) alias the exchange area
  static constexpr int SB_OFF = EX_OFF + 2 * EX_WIN;
  static_assert(4 * WT_BYTES <= 2 * EX_WIN, "staging tiles must fit the exchange area");
  static constexpr int SB_PITCH = 172;             // floats per head of the relative-position table (169 used)
  static constexpr int QB_OFF = SB_OFF + NH * SB_PITCH * 4;  // qkv bias [3C] floats (read per head with ds_read: no
  static constexpr int LDS = QB_OFF + 3 * C * 4;             // register global loads inside the DMA loop)
  static_assert(HSTEPS % G == 0 && SLOTS <= G, "ring geometry");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int C, bool SINGLE>  // SINGLE: the arithmetic mode as a compile-time parameter (no uniform branches around the lo-term MFMAs)
__global__ void __launch_bounds__(256, SwinCfg<C>::WG_PER_CU) swin_attn_fused_kernel(const SwinArgs p) {
  using K_ = SwinCfg<C>;
  constexpr int KS = K_::KS, NTL = K_::NTL, NH = K_::NH, HSTEPS = K_::HSTEPS, G = K_::G, STAGE = K_::STAGE, SLOTS = K_::SLOTS;
  constexpr int NP = C / 32;
  __shared__ __attribute__((aligned(16))) unsigned char smem[K_::LDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int it = wave >> 1, qt = wave & 1;  // window slot of the workgroup, query / key tile of the window
  const int l31 = lane & 31, hf = lane >> 5;
  constexpr int single = SINGLE;
  if (p.T < 0) reinterpret_cast<u32x4*>(smem)[tid] = u32x4{0u, 0u, 0u, 0u};  // a visible store (see ffn_fused_kernel)

  // ---- weight stream: this wave moves pieces wave, wave + 4, ... of every stage
  const unsigned char* wp = p.wpk + (long long)wave * PIECE;
  const unsigned voff = lane * 16;
  auto dma = [&](int stage_base, int q) {
    glds16(wp, voff, (unsigned)(stage_base + (q * 4 + wave) * PIECE));
    wp += 4 * PIECE;
  };
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) dma(0, q);

  // ---- the window and its tokens
  const long long win = (long long)blockIdx.x * 2 + it;
  const bool active = win < p.nwin;
  const int Hp = p.nWy * WS, Wp = p.nWx * WS;
  int wx = 0, wy = 0, t = 0;
  if (active) {
    long long r = win;
    wx = (int)(r % p.nWx); r /= p.nWx;
    wy = (int)(r % p.nWy); r /= p.nWy;
    t = (int)r;
  }
  // window token j -> row of x (frame t, un-shifted pixel), or -1 for a padded / non-existent token
  auto row_of = [&](int j) -> int {
    if (!active || j >= NTOK) return -1;
    int ys = wy * WS + j / WS + p.shift, xs = wx * WS + j % WS + p.shift;
    if (ys >= Hp) ys -= Hp;
    if (xs >= Wp) xs -= Wp;
    return (ys < p.H && xs < p.W) ? (t * p.H + ys) * p.W + xs : -1;
  };
  // coalesced side: lane (cr = lane >> 3, 16-byte piece lane & 7) touches rows 8i + cr of the wave's tile
  const int cr = lane >> 3, cp = (lane & 7) * 4;
  int rowc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rowc[i] = row_of(32 * qt + 8 * i + cr);

  // relative-position table of every head -> LDS ([head][169])
  float* const sB = reinterpret_cast<float*>(smem + K_::SB_OFF);
  for (int i = tid; i < NH * 169; i += 256) sB[(i / 169) * K_::SB_PITCH + i % 169] = p.table[(i % 169) * NH + i / 169];
  float* const sQB = reinterpret_cast<float*>(smem + K_::QB_OFF);
  for (int i = tid; i < 3 * C; i += 256) sQB[i] = p.bqkv[i];

  // ---- x rows of the tile -> norm1 -> fp16 hi/lo fragments (lane (token l31, hf) holds channels 16s + 8hf + 0..7); padded
  // tokens are ZERO after the norm (swin_transformer.py:214-222: F.pad follows norm1)
  float* const wt = reinterpret_cast<float*>(smem + K_::WT_OFF + wave * WT_BYTES);
  h16x8 xh[KS], xl[KS];
  {
    f32x4 v[NP][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* px = p.x + (long long)max(rowc[i], 0) * p.ldx + cp;
#pragma unroll
      for (int q = 0; q < NP; ++q) v[q][i] = rowc[i] >= 0 ? *reinterpret_cast<const f32x4*>(px + 32 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float mean[4], rstd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < NP; ++q) sum += (v[q][i][0] + v[q][i][1]) + (v[q][i][2] + v[q][i][3]);
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      sum += __shfl_xor(sum, 4, 64);
      mean[i] = sum * (1.f / C);
      float sq = 0.f;
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[q][i][e] - mean[i];
          sq = fmaf(d, d, sq);
        }
      sq += __shfl_xor(sq, 1, 64);
      sq += __shfl_xor(sq, 2, 64);
      sq += __shfl_xor(sq, 4, 64);
      rstd[i] = rowc[i] >= 0 ? rsqrtf(sq * (1.f / C) + p.eps) : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(p.g1 + 32 * q + cp);
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.be1 + 32 * q + cp);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[q][i][e] = rowc[i] >= 0 ? (v[q][i][e] - mean[i]) * rstd[i] * g[e] + b[e] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(wt + (8 * i + cr) * WT_PITCH + cp) = v[q][i];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const float* pr = wt + l31 * WT_PITCH + 16 * h2 + 8 * hf;
        const f32x4 a = *reinterpret_cast<const f32x4*>(pr);
        const f32x4 b = *reinterpret_cast<const f32x4*>(pr + 4);
        const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        const HL sp = split8(f, single);
        xh[2 * q + h2] = sp.hi;
        xl[2 * q + h2] = sp.lo;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }

  // ---- this lane's query token; per key slot (key tile kt, accumulator register r: key 32 kt + crow(r, hf)) the relative-
  // position index and the -100 shift mask come from tables the host built (no per-lane index arithmetic):
  //   index = (iy*13 + ix + 84) - (jy*13 + jx);  the mask separates regions of the padded, shifted grid (swin_transformer.py:
  //   370-388): only the last window row / column has two (rows < 7 - shift | rows >= 7 - shift), a pair is masked iff the
  //   query and the key lie on different sides in the row or in the column direction.
  const int qi = min(32 * qt + l31, NTOK - 1);
  const int iy = qi / WS, ix = qi % WS;
  const int qbase = iy * (2 * WS - 1) + ix + (WS - 1) * (2 * WS - 1) + (WS - 1);
  unsigned cj[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) cj[i] = hf ? p.cj[1][i] : p.cj[0][i];
  const unsigned kvalid = hf ? p.kvalid[1] : p.kvalid[0];
  unsigned kmask = 0u;
  if (p.shift > 0) {
    const unsigned rowhi = hf ? p.rowhi[1] : p.rowhi[0], colhi = hf ? p.colhi[1] : p.colhi[0];
    if (wy == p.nWy - 1) kmask |= (iy >= WS - p.shift) ? ~rowhi : rowhi;
    if (wx == p.nWx - 1) kmask |= (ix >= WS - p.shift) ? ~colhi : colhi;
  }

  f32x16 oacc[NTL];
#pragma unroll
  for (int tt = 0; tt < NTL; ++tt)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[tt][i] = 0.f;
  tce_amax_t amax = 0;
  unsigned char* const ex = smem + K_::EX_OFF + it * K_::EX_WIN;  // this window's K | V exchange area
  const float scale = 0.17677669529663687f;                     // 32^-0.5

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int stage = 0;  // ring stage (0 / 1) being consumed
  for (int h = 0; h < NH; ++h) {
    f32x16 acc;  // the projection being accumulated (q, then k, then v)
    HL qB[2];    // q^T of the head as B fragments
    h16x8 oh[2], ol[2];  // O^T of the head as B fragments (built after the attention core)
    static_for<0, HSTEPS>([&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      constexpr int sg = s % G;  // position inside the ring stage
      const unsigned char* const st = smem + stage * STAGE + lane * 16 + (2 * sg) * PIECE;
      const h16x8 fh = *reinterpret_cast<const h16x8*>(st);
      const h16x8 fl = *reinterpret_cast<const h16x8*>(st + PIECE);
      if constexpr (sg < SLOTS) dma((stage ^ 1) * STAGE, sg);  // the next stage's pieces, one DMA per step
      if constexpr (s < 3 * KS) {
        constexpr int ph = s / KS, ks = s % KS;  // 0 q, 1 k, 2 v
        if constexpr (ks == 0) {  // the accumulator starts from the bias: q / k rows in accumulator order, v per lane (d = l31)
          if constexpr (ph < 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 bb = *reinterpret_cast<const f32x4*>(sQB + ph * C + h * 32 + 8 * g + 4 * hf);
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[4 * g + c] = bb[c];
            }
          } else {
            const float bv = sQB[2 * C + h * 32 + l31];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = bv;
          }
        }
        if constexpr (ph < 2) mma3(acc, fh, fl, xh[ks], xl[ks], single);  // W x^T: lane = token
        else mma3(acc, xh[ks], xl[ks], fh, fl, single);                   // x W^T: lane = d
        if constexpr (ks == KS - 1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) amax = tce_amax1(amax, acc[i]);
          if constexpr (ph == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] *= scale;
            qB[0] = acc_frag(acc, 0, single);
            qB[1] = acc_frag(acc, 1, single);
          } else {  // K (ph 1) / V (ph 2) of this wave's key tile -> the window's exchange area
            unsigned char* const dst = ex + ((ph - 1) * 2 + qt) * 4 * PIECE + lane * 16;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const HL f = acc_frag(acc, s2, single);
              *reinterpret_cast<h16x8*>(dst + (2 * s2) * PIECE) = f.hi;
              *reinterpret_cast<h16x8*>(dst + (2 * s2 + 1) * PIECE) = f.lo;
            }
          }
        }
      }
      if constexpr (s == 3 * KS - 1) {
        // ---- attention core of the head: K and V of both key tiles are in LDS once every wave has passed this barrier
        __syncthreads();
        f32x16 sc[2];
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
          for (int i = 0; i < 16; ++i) sc[kt][i] = 0.f;
          const unsigned char* const kp = ex + (0 * 2 + kt) * 4 * PIECE + lane * 16;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
            mma3(sc[kt], *reinterpret_cast<const h16x8*>(kp + (2 * s2) * PIECE), *reinterpret_cast<const h16x8*>(kp + (2 * s2 + 1) * PIECE),
                 qB[s2].hi, qB[s2].lo, single);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int slot = kt * 16 + r;
            float a = sc[kt][r] + sB[h * K_::SB_PITCH + qbase - (int)((cj[slot >> 2] >> (8 * (slot & 3))) & 0xffu)];
            if (kmask & (1u << slot)) a += -100.0f;
            a = (kvalid & (1u << slot)) ? a : -3.0e38f;
            sc[kt][r] = a;
            mx = fmaxf(mx, a);
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pj = sc[kt][r] > -1.0e38f ? __expf(sc[kt][r] - mx) : 0.f;
            sc[kt][r] = pj;
            l += pj;
          }
        l += __shfl_xor(l, 32, 64);
        f32x16 o;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const unsigned char* const vp = ex + (1 * 2 + kt) * 4 * PIECE + lane * 16;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const HL pf = acc_frag(sc[kt], s2, single);
            mma3(o, *reinterpret_cast<const h16x8*>(vp + (2 * s2) * PIECE), *reinterpret_cast<const h16x8*>(vp + (2 * s2 + 1) * PIECE),
                 pf.hi, pf.lo, single);
          }
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          o[i] *= inv;
          amax = tce_amax1(amax, o[i]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const HL f = acc_frag(o, s2, single);
          oh[s2] = f.hi;
          ol[s2] = f.lo;
        }
      }
      if constexpr (s >= 3 * KS) {  // output projection: out^T[32 tt .., tok] += Wp[32 tt .., head's d] O^T
        constexpr int j = s - 3 * KS, tt = j >> 1, s2 = j & 1;
        mma3(oacc[tt], fh, fl, oh[s2], ol[s2], single);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (sg == G - 1) {  // end of a ring stage: the next stage has landed, every wave is done reading this one
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stage ^= 1;
      }
    });
  }

  // ---- epilogue: + proj bias + residual (the un-normed rows again), stores; full 128-byte lines through the staging tile
#pragma unroll
  for (int tt = 0; tt < NTL; ++tt) {
    f32x4 rv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      rv[i] = rowc[i] >= 0 ? *reinterpret_cast<const f32x4*>(p.x + (long long)rowc[i] * p.ldx + 32 * tt + cp) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bp = *reinterpret_cast<const f32x4*>(p.bproj + 32 * tt + 8 * g + 4 * hf);
      const f32x4 o = {oacc[tt][4 * g] + bp[0], oacc[tt][4 * g + 1] + bp[1], oacc[tt][4 * g + 2] + bp[2], oacc[tt][4 * g + 3] + bp[3]};
      *reinterpret_cast<f32x4*>(wt + l31 * WT_PITCH + 8 * g + 4 * hf) = o;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 o = *reinterpret_cast<const f32x4*>(wt + (8 * i + cr) * WT_PITCH + cp);
      o += rv[i];
      if (rowc[i] >= 0) {
        amax = tce_amax4(amax, o);
        *reinterpret_cast<f32x4*>(p.out + (long long)rowc[i] * p.ldo + 32 * tt + cp) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  tce_range_report(p.range_flag, amax);
}

// One thread per 16-byte unit of the packed stream: per head HSTEPS (hi, lo) piece pairs in consumption order
// [Wq_h: KS steps | Wk_h | Wv_h | Wp[:, head] : NTL tiles x 2 k-steps], then one stage of zero padding at the very end
// (the last stage's prefetch reads defined bytes that are never consumed).
__global__ void __launch_bounds__(256) swin_attn_pack_kernel(const float* __restrict__ Wqkv, const float* __restrict__ Wp,
                                                             unsigned char* __restrict__ out, const int C, const long long units,
                                                             const long long real_units, const int single) {
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  if (u >= units) return;
  u32x4 o = {0u, 0u, 0u, 0u};
  if (u < real_units) {
    const int KS = C / 16, HSTEPS = C / 4;
    const int lane = (int)(u & 63);
    const long long pg = u >> 6;
    const int head = (int)(pg / (2 * HSTEPS)), within = (int)(pg % (2 * HSTEPS));
    const int step = within >> 1, plane = within & 1;
    const int r = lane & 31, hf = lane >> 5;
    float v[8];
    if (step < 3 * KS) {
      const int ph = step / KS, ks = step % KS;
      const float* row = Wqkv + (long long)(ph * C + head * 32 + r) * C + 16 * ks + 8 * hf;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = row[j];
    } else {
      const int j2 = step - 3 * KS, tt = j2 >> 1, s2 = j2 & 1;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = Wp[(long long)(32 * tt + r) * C + head * 32 + 16 * s2 + 8 * (j >> 2) + 4 * hf + (j & 3)];
    }
    const HL f = split8(v, single);
    o = __builtin_bit_cast(u32x4, plane ? f.lo : f.hi);
  }
  reinterpret_cast<u32x4*>(out)[u] = o;
}

bool swin_shape_ok(int C) { return C == 96 || C == 128 || C == 192 || C == 256; }
long long swin_real_units(int C) { return (long long)(C / 32) * 2 * (C / 4) * 64; }
long long swin_units(int C) { return swin_real_units(C) + (long long)((C <= SWIN_OCC2_MAXC) ? 8 : 16) * 2 * 64; }

template <int C>
void swin_launch(const SwinArgs& a, hipStream_t s) {
  if (a.single) hipLaunchKernelGGL((swin_attn_fused_kernel<C, true>), dim3(tce_cdiv(a.nwin, 2)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((swin_attn_fused_kernel<C, false>), dim3(tce_cdiv(a.nwin, 2)), dim3(256), 0, s, a);
}

}  // namespace

extern "C" int64_t tce_swin_attn_packed_bytes(int32_t C) { return swin_shape_ok(C) ? swin_units(C) * 16 : -1; }

extern "C" int tce_swin_attn_pack_f32(const float* Wqkv, const float* Wproj, void* packed, int32_t C, tceStream stream) {
  TCE_CHECK_ARG(swin_shape_ok(C), "tce_swin_attn_pack_f32: unsupported C=%d (96, 128, 192, 256)", C);
  TCE_CHECK_ARG(Wqkv && Wproj && packed && tce_aligned16(packed), "tce_swin_attn_pack_f32: null / misaligned pointer");
  const long long units = swin_units(C);
  hipLaunchKernelGGL(swin_attn_pack_kernel, dim3(tce_cdiv(units, 256)), dim3(256), 0, (hipStream_t)stream, Wqkv, Wproj,
                     (unsigned char*)packed, C, units, swin_real_units(C), tce_gemm_single_pass());
  TCE_CHECK_LAUNCH("tce_swin_attn_pack_f32");
  return TCE_OK;
}

extern "C" int tce_swin_attn_fused_f32(const float* x, int64_t ldx, const void* packed, const float* qkv_bias, const float* proj_bias,
                                       const float* bias_table, const float* gamma1, const float* beta1, float eps, float* out,
                                       int64_t ldo, int32_t T, int32_t H, int32_t W, int32_t C, int32_t shift, tceStream stream) {
  TCE_CHECK_ARG(swin_shape_ok(C), "tce_swin_attn_fused_f32: unsupported C=%d (96, 128, 192, 256)", C);
  TCE_CHECK_ARG(x && packed && qkv_bias && proj_bias && bias_table && gamma1 && beta1 && out, "tce_swin_attn_fused_f32: null pointer");
  TCE_CHECK_ARG(T > 0 && H > 0 && W > 0 && (long long)T * H * W < (1ll << 31) && shift >= 0 && shift < WS,
                "tce_swin_attn_fused_f32: bad sizes / shift");
  TCE_CHECK_ARG(ldx >= C && ldx % 4 == 0 && ldo >= C && ldo % 4 == 0, "tce_swin_attn_fused_f32: bad row pitch");
  TCE_CHECK_ARG(tce_aligned16(x) && tce_aligned16(out) && tce_aligned16(packed) && tce_aligned16(qkv_bias) && tce_aligned16(proj_bias) &&
                    tce_aligned16(gamma1) && tce_aligned16(beta1),
                "tce_swin_attn_fused_f32: pointers must be 16-byte aligned");
  TCE_CHECK_ARG(out == x || out + (long long)T * H * W * ldo <= x || x + (long long)T * H * W * ldx <= out,
                "tce_swin_attn_fused_f32: out must be x (in place) or not overlap it");
  SwinArgs a;
  a.x = x; a.out = out; a.wpk = (const unsigned char*)packed; a.bqkv = qkv_bias; a.bproj = proj_bias; a.table = bias_table;
  a.g1 = gamma1; a.be1 = beta1; a.ldx = ldx; a.ldo = ldo; a.T = T; a.H = H; a.W = W;
  a.nWy = (H + WS - 1) / WS; a.nWx = (W + WS - 1) / WS; a.shift = shift;
  a.nwin = (long long)T * a.nWy * a.nWx; a.eps = eps; a.range_flag = tce_range_flag(); a.single = tce_gemm_single_pass();
  for (int hf = 0; hf < 2; ++hf) {
    a.kvalid[hf] = a.rowhi[hf] = a.colhi[hf] = 0u;
    for (int i = 0; i < 8; ++i) a.cj[hf][i] = 0u;
    for (int slot = 0; slot < 32; ++slot) {
      const int kt = slot >> 4, r = slot & 15;
      const int j = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf, jc = j < NTOK ? j : NTOK - 1;
      const int jy = jc / WS, jx = jc % WS;
      a.cj[hf][slot >> 2] |= (unsigned)(jy * (2 * WS - 1) + jx) << (8 * (slot & 3));
      if (j < NTOK) a.kvalid[hf] |= 1u << slot;
      if (jy >= WS - shift) a.rowhi[hf] |= 1u << slot;
      if (jx >= WS - shift) a.colhi[hf] |= 1u << slot;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  if (C == 256) swin_launch<256>(a, s);
  else if (C == 192) swin_launch<192>(a, s);
  else if (C == 128) swin_launch<128>(a, s);
  else swin_launch<96>(a, s);
  TCE_CHECK_LAUNCH("tce_swin_attn_fused_f32");
  return TCE_OK;
}
