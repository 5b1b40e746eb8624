// Token-stationary fused FFN on the fp16 matrix cores (3 x fp16 split, fp32 accumulate):
//
//     out = LN_out?( x + W2 act(W1 LN_in?(x) + b1) + b2 )
//
// Replaces linear1 -> ReLU -> linear2 -> +residual -> LayerNorm of the deformable-transformer / VisionLanguageBlock
// FFNs (tce_deformable_transformer.py:489-491,548-552; segmentation.py:374-376) and LayerNorm -> fc1 -> GELU -> fc2 ->
// +residual of the Swin MLP (swin_transformer.py:28-47,255-256) by ONE launch in which the [M, hidden] tensor never
// exists: it lives 32 hidden units at a time in the accumulator registers of the wave that owns the token.
//
// Dataflow ("everything transposed, the lane owns a token").  A wave owns 32 tokens = the 32 columns of every MFMA it
// issues (v_mfma_f32_32x32x16_f16; D = A B + C with A rows in registers-of-D, B columns on lanes):
//   * x[32 tokens, C] is loaded ONCE, straight from HBM into registers in B-fragment order, split into fp16 hi/lo
//     (hi = f16_rtz(x), lo = f16_rtz(x - hi)) and stays there for the whole kernel (C/2 registers);
//   * chunk c (32 hidden units):  H^T[32 hidden, 32 tokens] = W1[c] x^T  -- A = W1 fragments from LDS, B = x registers;
//     bias enters as the accumulator's initial value; activation and the hi/lo split run on the accumulator registers;
//   * out^T[C, 32 tokens] += W2[:, c] H^T -- A = W2 fragments from LDS, B = the SAME registers: a 32x32 accumulator
//     tile has its column (token) on the lane and its rows (hidden) in the 16 registers, which is exactly a
//     B operand that sums over the row index; no lane movement, no LDS.  Registers 8s..8s+7 form k-step s, in the
//     order k = 16s + 8(j>>2) + 4(lane>>5) + (j&3); the W2 fragments are packed in that same order;
//   * epilogue on the out^T accumulators (lane = token, registers = 4-channel groups): + b2 + residual, optional
//     LayerNorm over the token's C channels (half of them in this lane, half in lane^32), 16-byte stores.
// Weights are static, so they are pre-split into fp16 hi/lo planes and pre-ordered at pack time
// (tce_ffn_pack_f32) into the exact sequence of 1 KiB MFMA A-fragments ("pieces": lane l's 16 bytes at l*16) the
// loop consumes.  One iteration's pieces = [b1 chunk i | W1 chunk i | W2 chunk i-1] move L2 -> LDS by DMA
// (global_load_lds_dwordx4, no VGPR staging, no ds_write, contiguous 1 KiB per wave-instruction) into a two-stage
// ring; every ds_read_b128 is `base + lane*16 + immediate`, conflict-free by construction.  W2 lags one chunk so the
// activation/split of chunk i overlaps the MFMAs of chunk i-1's second product (iteration 0 and the last iteration
// see zero blocks).  One workgroup barrier per iteration (96 MFMAs per wave at C = 256).
//
// Register budget at C = 256: x 128 + out 128 + H 16 + fragments ~48 -> one wave per SIMD (4 waves, 512 registers);
// C <= 128 runs 8 waves (two per SIMD).  Per-wave LDS traffic: 64 ds_read_b128 per 96 MFMAs; per-CU fill:
// 65 KiB per iteration.  Algorithmic HBM bytes: x read + out written (2 * M * C * 4) + the weight stream once per XCD.
#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/tce_rvos.h"
#include "../../include/tce_rvos_debug.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PIECE = 1024;  // one wave-wide 16-byte-per-lane DMA / ds_read_b128

// Ablation builds (tools/ffn_ablate.py compiles this file stand-alone with -DFFN_ABL=n; results are then wrong):
// bit 0: no DMA inside the loop; bit 1: no vmcnt wait / barrier inside the loop; bit 2: no MFMA (fragments kept live).
#ifndef FFN_ABL
#define FFN_ABL 0
#endif
// Diagnostic stamps (tce_debug_ffn_set_stamp_buffer): lane 0 of wave 0 of the first 1024 workgroups records
// {s_memtime at entry, after the prologue, after the loop, at exit, s_memrealtime at entry, at exit}.
static __device__ long long* g_ffn_stamps = nullptr;

// One DMA piece: 64 lanes x 16 bytes from sbase + voff (voff = lane*16) to LDS byte offset lds_dst (+ lane*16, added
// by the hardware).  M0 is not restored: the compiler re-materialises M0 immediately before each of its own uses and
// this file's kernels have none (checked in the ISA: only these statements touch m0).
__device__ __forceinline__ void glds16(const unsigned char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_dst)
               : "memory");
}

struct HL {
  h16x8 hi, lo;
};

// 8 fp32 -> fp16 hi (truncated) + fp16 lo (truncated exact remainder); element j of the fragment = v[j]
__device__ __forceinline__ HL split8(const float* v) {
  u32x4 h, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const fp16x2_t a = __builtin_amdgcn_cvt_pkrtz(v[2 * q], v[2 * q + 1]);
    const fp16x2_t b = __builtin_amdgcn_cvt_pkrtz(v[2 * q] - (float)a[0], v[2 * q + 1] - (float)a[1]);
    h[q] = __builtin_bit_cast(unsigned, a);
    l[q] = __builtin_bit_cast(unsigned, b);
  }
  HL r;
  r.hi = __builtin_bit_cast(h16x8, h);
  r.lo = __builtin_bit_cast(h16x8, l);
  return r;
}

struct FfnArgs {
  const float* x;
  const unsigned char* wpk;
  const float* b2;
  const float *g_in, *be_in, *g_out, *be_out;
  float* out;
  long long ldx, ldo;
  int M, NI;  // NI = hidden/32 + 1 iterations
  float eps_in, eps_out;
  int* range_flag;  // tce_set_range_flag: set when a hidden or an output value leaves the fp16 range of the split
};

template <int C, int WAVES, int ACT>
__global__ void __launch_bounds__(64 * WAVES, WAVES / 4) ffn_fused_kernel(const FfnArgs p) {
  constexpr int KS = C / 16, NT = C / 32;
  constexpr int SLOTS = (1 + 2 * KS + 4 * NT + WAVES - 1) / WAVES;
  constexpr int P = SLOTS * WAVES;  // pieces per stage, padded so that every wave issues exactly SLOTS DMAs (no branches)
  constexpr int STAGE = P * PIECE;
  constexpr int STEPS = KS + 2 * NT;
  static_assert(SLOTS <= STEPS, "more DMA slots than loop steps");
  static_assert(2 * STAGE <= 160 * 1024, "ring does not fit the LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];  // the ONLY LDS object: base offset 0

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hf = lane >> 5;
  const int m = blockIdx.x * (32 * WAVES) + wave * 32 + r;
  const int mc = min(m, p.M - 1);

  // The ring is written only by DMA (inline asm), which the compiler cannot see: without a visible store it treats
  // every read of `smem` as undefined and deletes it.  This store never executes (NI >= 2 always).
  if (p.NI < 0) reinterpret_cast<u32x4*>(smem)[tid] = u32x4{0u, 0u, 0u, 0u};

  long long* const stamps = (g_ffn_stamps && blockIdx.x < 1024 && tid == 0) ? g_ffn_stamps + blockIdx.x * 8 : nullptr;
  if (stamps) {
    stamps[0] = (long long)__builtin_amdgcn_s_memtime();
    stamps[4] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  // this wave's pieces are wave, wave + WAVES, ... of every block, and blocks are contiguous: the source pointer
  // simply advances by WAVES pieces per DMA for the whole kernel (the stream ends with one block of padding, so the
  // last iteration's prefetch reads defined bytes that are never consumed)
  const unsigned char* wp = p.wpk + (long long)wave * PIECE;
  const unsigned voff = lane * 16;
  const unsigned wbase = wave * PIECE;
  auto dma = [&](int stage, int q) {
    glds16(wp, voff, wbase + (unsigned)(stage * STAGE + q * WAVES * PIECE));
    wp += WAVES * PIECE;
  };
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) dma(0, q);

  // ---- x: 32 tokens x C, B-fragment order (lane (r, hf) holds k = 16s + 8hf + 0..7 of token r), optional LayerNorm
  h16x8 xh[KS], xl[KS];
  {
    float xf[KS][8];
    const float* px = p.x + (long long)mc * p.ldx + 8 * hf;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(px + 16 * s);
      const f32x4 b = *reinterpret_cast<const f32x4*>(px + 16 * s + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xf[s][j] = a[j];
        xf[s][4 + j] = b[j];
      }
    }
    if (p.g_in) {
      float sum = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += xf[s][j];
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * (1.f / C);
      float sq = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = xf[s][j] - mean;
          sq += d * d;
        }
      sq += __shfl_xor(sq, 32, 64);
      const float rstd = rsqrtf(sq * (1.f / C) + p.eps_in);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.g_in + 16 * s + 8 * hf);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(p.g_in + 16 * s + 8 * hf + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.be_in + 16 * s + 8 * hf);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.be_in + 16 * s + 8 * hf + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xf[s][j] = (xf[s][j] - mean) * rstd * g0[j] + b0[j];
          xf[s][4 + j] = (xf[s][4 + j] - mean) * rstd * g1[j] + b1[j];
        }
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const HL f = split8(xf[s]);
      xh[s] = f.hi;
      xl[s] = f.lo;
    }
  }

  f32x16 oacc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[t][i] = 0.f;
  tce_amax_t amax = 0;
  h16x8 hh0, hl0, hh1, hl1;  // H^T of the previous chunk as B fragments (k-steps 0 and 1)
#pragma unroll
  for (int j = 0; j < 8; ++j) hh0[j] = hl0[j] = hh1[j] = hl1[j] = (_Float16)0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // piece index of step `st` (0..STEPS-1) inside a stage: hi piece; the lo piece follows it
  auto piece_of = [](int st) { return 1 + 2 * st; };
  auto body = [&](auto stage_c) {
    constexpr int SB = decltype(stage_c)::value * STAGE;
    const unsigned char* const st = smem + SB + lane * 16;
    f32x16 hacc;
    {
      const f32x4* bp = reinterpret_cast<const f32x4*>(smem + SB + hf * 64);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = bp[g];
#pragma unroll
        for (int c = 0; c < 4; ++c) hacc[4 * g + c] = b[c];
      }
    }
    u32x4 nh[2], nl[2];  // this chunk's H^T as B fragments, built pair by pair between the second product's MFMAs
    h16x8 ah = *reinterpret_cast<const h16x8*>(st + piece_of(0) * PIECE);
    h16x8 al = *reinterpret_cast<const h16x8*>(st + (piece_of(0) + 1) * PIECE);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      h16x8 fh = ah, fl = al;
      if (s + 1 < STEPS) {
        fh = *reinterpret_cast<const h16x8*>(st + piece_of(s + 1) * PIECE);
        fl = *reinterpret_cast<const h16x8*>(st + (piece_of(s + 1) + 1) * PIECE);
      }
      if (s < SLOTS && !(FFN_ABL & 1)) dma(decltype(stage_c)::value ^ 1, s);
      if (FFN_ABL & 4) {
        asm volatile("" ::"v"(ah), "v"(al));
      } else if (s < KS) {
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[s], hacc, 0, 0, 0);
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[s], hacc, 0, 0, 0);
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[s], hacc, 0, 0, 0);
      } else {
        const int j = s - KS, t = j >> 1;
        if (j & 1) {
          oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, hl1, oacc[t], 0, 0, 0);
          oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, hh1, oacc[t], 0, 0, 0);
          oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, hh1, oacc[t], 0, 0, 0);
        } else {
          oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, hl0, oacc[t], 0, 0, 0);
          oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, hh0, oacc[t], 0, 0, 0);
          oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, hh0, oacc[t], 0, 0, 0);
        }
        // activation + hi/lo split of value pairs [8j/(2NT), 8(j+1)/(2NT)) of this chunk, spread over the 2NT steps
#pragma unroll
        for (int q = (8 * j) / (2 * NT); q < (8 * (j + 1)) / (2 * NT); ++q) {
          float v0 = hacc[2 * q], v1 = hacc[2 * q + 1];
          if (ACT == 1) {  // plain v_max: fmaxf() would first quiet a possible sNaN with a second v_max per value
            asm("v_max_f32_e32 %0, 0, %1" : "=v"(v0) : "v"(v0));
            asm("v_max_f32_e32 %0, 0, %1" : "=v"(v1) : "v"(v1));
          }
          if (ACT == 2) {
            v0 = 0.5f * v0 * (1.f + erff(v0 * 0.70710678118654752440f));
            v1 = 0.5f * v1 * (1.f + erff(v1 * 0.70710678118654752440f));
          }
          amax = max(amax, max(tce_absbits(v0), tce_absbits(v1)));
          const fp16x2_t a = __builtin_amdgcn_cvt_pkrtz(v0, v1);
          const fp16x2_t b = __builtin_amdgcn_cvt_pkrtz(v0 - (float)a[0], v1 - (float)a[1]);
          unsigned wa = __builtin_bit_cast(unsigned, a), wb = __builtin_bit_cast(unsigned, b);
          asm volatile("" : "+v"(wa), "+v"(wb));  // anchors the computation in this step (the optimiser would sink it)
          nh[q >> 2][q & 3] = wa;
          nl[q >> 2][q & 3] = wb;
        }
      }
      ah = fh;
      al = fl;
      // pin the step: the compiler would otherwise sink the whole activation/split block behind the last MFMA of
      // the iteration (its results are only needed an iteration later), where nothing hides it
      __builtin_amdgcn_sched_barrier(0);
    }
    hh0 = __builtin_bit_cast(h16x8, nh[0]);
    hl0 = __builtin_bit_cast(h16x8, nl[0]);
    hh1 = __builtin_bit_cast(h16x8, nh[1]);
    hl1 = __builtin_bit_cast(h16x8, nl[1]);
    if (!(FFN_ABL & 2)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  };
  if (stamps) stamps[1] = (long long)__builtin_amdgcn_s_memtime();
  int it = 0;
  for (; it + 1 < p.NI; it += 2) {
    body(std::integral_constant<int, 0>{});
    body(std::integral_constant<int, 1>{});
  }
  if (it < p.NI) body(std::integral_constant<int, 0>{});
  if (FFN_ABL & 3) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (stamps) stamps[2] = (long long)__builtin_amdgcn_s_memtime();

  // ---- epilogue: lane = token, accumulator registers 4g..4g+3 of tile t = channels 32t + 8g + 4hf + (0..3)
  const float* xr = p.x + (long long)mc * p.ldx + 4 * hf;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(p.b2 + 32 * t + 8 * g + 4 * hf);
      const f32x4 rv = *reinterpret_cast<const f32x4*>(xr + 32 * t + 8 * g);
#pragma unroll
      for (int c = 0; c < 4; ++c) oacc[t][4 * g + c] += bv[c] + rv[c];
    }
  if (p.g_out) {
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) sum += oacc[t][i];
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.f / C);
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float d = oacc[t][i] - mean;
        sq += d * d;
      }
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = rsqrtf(sq * (1.f / C) + p.eps_out);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(p.g_out + 32 * t + 8 * g + 4 * hf);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.be_out + 32 * t + 8 * g + 4 * hf);
#pragma unroll
        for (int c = 0; c < 4; ++c) oacc[t][4 * g + c] = (oacc[t][4 * g + c] - mean) * rstd * gv[c] + bv[c];
      }
  }
  if (m < p.M) {
    float* po = p.out + (long long)m * p.ldo + 4 * hf;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = oacc[t][4 * g + c];
        amax = tce_amax4(amax, o);
        *reinterpret_cast<f32x4*>(po + 32 * t + 8 * g) = o;
      }
  }
  tce_range_report(p.range_flag, amax);
  if (stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamps[3] = (long long)__builtin_amdgcn_s_memtime();
    stamps[5] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}

// One thread per 16-byte unit of the packed stream (see the header of this file for the order).
__global__ void __launch_bounds__(256) ffn_pack_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                                       const float* __restrict__ W2, unsigned char* __restrict__ out,
                                                       const int C, const int Hd, const int P,
                                                       const long long units) {
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  if (u >= units) return;
  const int KS = C / 16, NT = C / 32, NC = Hd / 32;
  const int lane = (int)(u & 63);
  const long long pg = u >> 6;
  const int piece = (int)(pg % P), it = (int)(pg / P);
  const int r = lane & 31, hf = lane >> 5;
  u32x4 o = {0u, 0u, 0u, 0u};
  if (piece == 0) {  // bias: [hf][16] floats, value[hf][i] = b1[32 it + (i&3) + 8(i>>2) + 4hf]
    if (lane < 8 && it < NC) {
      f32x4 v;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int idx = 4 * lane + c, h2 = idx >> 4, i = idx & 15;
        v[c] = b1 ? b1[32 * it + (i & 3) + 8 * (i >> 2) + 4 * h2] : 0.f;
      }
      o = __builtin_bit_cast(u32x4, v);
    }
  } else if (piece < 1 + 2 * KS + 4 * NT) {
    float v[8];
    int plane;
    if (piece <= 2 * KS) {
      const int s = (piece - 1) >> 1;
      plane = (piece - 1) & 1;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = it < NC ? W1[(long long)(32 * it + r) * C + 16 * s + 8 * hf + j] : 0.f;
    } else {
      const int q = piece - 1 - 2 * KS;
      plane = q & 1;
      const int idx = q >> 1, t = idx >> 1, s2 = idx & 1, chunk = it - 1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 32 * chunk + 16 * s2 + 8 * (j >> 2) + 4 * hf + (j & 3);
        v[j] = (chunk >= 0 && chunk < NC) ? W2[(long long)(32 * t + r) * Hd + k] : 0.f;
      }
    }
    const HL f = split8(v);
    o = __builtin_bit_cast(u32x4, plane ? f.lo : f.hi);
  }
  reinterpret_cast<u32x4*>(out)[u] = o;
}

inline bool ffn_shape_ok(int C, int Hd) { return (C == 96 || C == 128 || C == 192 || C == 256) && Hd > 0 && Hd % 32 == 0; }
inline int ffn_waves(int C) { return C <= 128 ? 8 : 4; }
inline int ffn_pieces(int C) {
  const int w = ffn_waves(C);
  return (1 + 2 * (C / 16) + 4 * (C / 32) + w - 1) / w * w;
}
inline long long ffn_units(int C, int Hd) { return (long long)(Hd / 32 + 2) * ffn_pieces(C) * 64; }  // + 1 block of padding

template <int C, int WAVES>
void ffn_launch(const FfnArgs& a, int act, hipStream_t s) {
  const dim3 grid(tce_cdiv(a.M, 32 * WAVES)), block(64 * WAVES);
  if (act == 1) hipLaunchKernelGGL((ffn_fused_kernel<C, WAVES, 1>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((ffn_fused_kernel<C, WAVES, 2>), grid, block, 0, s, a);
}

}  // namespace

extern "C" int tce_debug_ffn_set_stamp_buffer(long long* dev_buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_ffn_stamps), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : -1;
}

extern "C" int64_t tce_ffn_packed_bytes(int32_t C, int32_t Hd) { return ffn_shape_ok(C, Hd) ? ffn_units(C, Hd) * 16 : -1; }

extern "C" int tce_ffn_pack_f32(const float* W1, const float* b1, const float* W2, void* packed, int32_t C, int32_t Hd,
                                tceStream stream) {
  TCE_CHECK_ARG(ffn_shape_ok(C, Hd), "tce_ffn_pack_f32: unsupported shape C=%d hidden=%d (C in 96/128/192/256, hidden %% 32 == 0)",
                C, Hd);
  TCE_CHECK_ARG(W1 && W2 && packed && tce_aligned16(packed), "tce_ffn_pack_f32: null / misaligned pointer");
  const long long units = ffn_units(C, Hd);
  hipLaunchKernelGGL(ffn_pack_kernel, dim3(tce_cdiv(units, 256)), dim3(256), 0, (hipStream_t)stream, W1, b1, W2,
                     (unsigned char*)packed, C, Hd, ffn_pieces(C), units);
  TCE_CHECK_LAUNCH("tce_ffn_pack_f32");
  return TCE_OK;
}

extern "C" int tce_ffn_fused_f32(const float* x, int64_t ldx, const void* packed, const float* b2, const float* g_in,
                                 const float* be_in, float eps_in, const float* g_out, const float* be_out,
                                 float eps_out, float* out, int64_t ldo, int32_t M, int32_t C, int32_t Hd, int32_t act,
                                 tceStream stream) {
  TCE_CHECK_ARG(ffn_shape_ok(C, Hd), "tce_ffn_fused_f32: unsupported shape C=%d hidden=%d", C, Hd);
  TCE_CHECK_ARG(M > 0 && x && packed && b2 && out, "tce_ffn_fused_f32: null pointer or M <= 0");
  TCE_CHECK_ARG(act == 1 || act == 2, "tce_ffn_fused_f32: act must be 1 (ReLU) or 2 (GELU)");
  TCE_CHECK_ARG(ldx >= C && ldo >= C && ldx % 4 == 0 && ldo % 4 == 0, "tce_ffn_fused_f32: bad row pitch");
  TCE_CHECK_ARG(tce_aligned16(x) && tce_aligned16(out) && tce_aligned16(packed) && tce_aligned16(b2),
                "tce_ffn_fused_f32: x/out/packed/b2 must be 16-byte aligned");
  TCE_CHECK_ARG((!g_in || (be_in && tce_aligned16(g_in) && tce_aligned16(be_in))) &&
                    (!g_out || (be_out && tce_aligned16(g_out) && tce_aligned16(be_out))),
                "tce_ffn_fused_f32: LayerNorm gamma/beta must come in pairs, 16-byte aligned");
  FfnArgs a;
  a.x = x; a.wpk = (const unsigned char*)packed; a.b2 = b2;
  a.g_in = g_in; a.be_in = be_in; a.g_out = g_out; a.be_out = be_out;
  a.out = out; a.ldx = ldx; a.ldo = ldo; a.M = M; a.NI = Hd / 32 + 1; a.eps_in = eps_in; a.eps_out = eps_out; a.range_flag = tce_range_flag();
  hipStream_t s = (hipStream_t)stream;
  if (C == 256) ffn_launch<256, 4>(a, act, s);
  else if (C == 192) ffn_launch<192, 4>(a, act, s);
  else if (C == 128) ffn_launch<128, 8>(a, act, s);
  else ffn_launch<96, 8>(a, act, s);
  TCE_CHECK_LAUNCH("tce_ffn_fused_f32");
  return TCE_OK;
}
