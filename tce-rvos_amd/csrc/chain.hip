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
#include "frag.h"
#include "../../include/tce_rvos.h"
#include "../../include/tce_rvos_debug.h"

namespace {

// Ablation builds (tools/ffn_ablate.py compiles this file stand-alone with -DFFN_ABL=n; results are then wrong):
// bit 0: no DMA inside the loop; bit 1: no vmcnt wait / barrier inside the loop; bit 2: no MFMA (fragments kept live).
#ifndef FFN_ABL
#define FFN_ABL 0
#endif
// Diagnostic stamps (tce_debug_ffn_set_stamp_buffer): lane 0 of wave 0 of the first 1024 workgroups records
// {s_memtime at entry, after the prologue, after the loop, at exit, s_memrealtime at entry, at exit}.
static __device__ long long* g_ffn_stamps = nullptr;

// ---------------------------------------------------------------------------------------------------------------
// Per-wave staging tile in LDS: 32 rows x 32 floats, 144-byte pitch.  Global memory is touched only in full 128-byte
// lines (one instruction = 8 rows x 128 B: lane l -> row 8i + (l>>3), 16-byte piece l&7), the MFMA side reads / writes
// row-per-lane; with the 144-byte pitch both patterns are bank-conflict free.  A wave's LDS operations execute in
// order, so its private tile needs no workgroup barrier -- only the compiler must not reorder (wave_barrier).
// ---------------------------------------------------------------------------------------------------------------

// x[32 tokens, K] (+ a2, LayerNorm) -> fp16 hi/lo B fragments (lane (r, hf) holds k = 16s + 8hf + 0..7 of token r).
// a2: optional addend with row pitch lda2; a2_rows > 0: its row index is (token % a2_rows) (a position map shared by
// all frames).  g_in / be_in: optional LayerNorm over K applied to (x + a2).
template <int K, bool HAS_A2 = true>
__device__ __forceinline__ void load_x_frags(const float* __restrict__ x, const long long ldx,
                                             const float* __restrict__ a2, const long long lda2, const int a2_rows,
                                             const int m0, const int M, float* __restrict__ wt, const int lane,
                                             const float* __restrict__ g_in, const float* __restrict__ be_in,
                                             const float eps, h16x8 (&xh)[K / 16], h16x8 (&xl)[K / 16],
                                             const int single = 0) {
  constexpr int NP = K / 32;
  const int cr = lane >> 3, cp = (lane & 7) * 4;
  const int r = lane & 31, hf = lane >> 5;
  f32x4 v[NP][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = min(m0 + 8 * i + cr, M - 1);
    const float* px = x + (long long)row * ldx + cp;
#pragma unroll
    for (int q = 0; q < NP; ++q) v[q][i] = *reinterpret_cast<const f32x4*>(px + 32 * q);
    if (HAS_A2 && a2) {
      const float* pa = a2 + (long long)(a2_rows > 0 ? row % a2_rows : row) * lda2 + cp;
#pragma unroll
      for (int q = 0; q < NP; ++q) v[q][i] += *reinterpret_cast<const f32x4*>(pa + 32 * q);
    }
  }
  if (g_in) {
    float mean[4], rstd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < NP; ++q) sum += (v[q][i][0] + v[q][i][1]) + (v[q][i][2] + v[q][i][3]);
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      sum += __shfl_xor(sum, 4, 64);
      mean[i] = sum * (1.f / K);
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
      rstd[i] = rsqrtf(sq * (1.f / K) + eps);
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(g_in + 32 * q + cp);
      const f32x4 b = *reinterpret_cast<const f32x4*>(be_in + 32 * q + cp);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[q][i][e] = (v[q][i][e] - mean[i]) * rstd[i] * g[e] + b[e];
    }
  }
#pragma unroll
  for (int q = 0; q < NP; ++q) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(wt + (8 * i + cr) * WT_PITCH + cp) = v[q][i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* pr = wt + r * WT_PITCH + 16 * h + 8 * hf;
      const f32x4 a = *reinterpret_cast<const f32x4*>(pr);
      const f32x4 b = *reinterpret_cast<const f32x4*>(pr + 4);
      const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
      const HL sp = split8(f, single);
      xh[2 * q + h] = sp.hi;
      xl[2 * q + h] = sp.lo;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// residual tile res[32 tokens, col0 .. col0+31] as four full-line loads per lane (issue them for ALL tiles first: a
// load -> LDS -> registers chain per tile would pay one memory round trip per tile)
struct ResTile {
  f32x4 v[4];
};
__device__ __forceinline__ ResTile tile_res_load(const float* __restrict__ res, const long long ldres, const int m0,
                                                 const int M, const int col0, const int lane) {
  const int cr = lane >> 3, cp = (lane & 7) * 4;
  ResTile t;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = min(m0 + 8 * i + cr, M - 1);
    t.v[i] = *reinterpret_cast<const f32x4*>(res + (long long)row * ldres + col0 + cp);
  }
  return t;
}

// bias[col0 .. col0+31] in accumulator order (4 x float4 per lane); loaded ahead of use like the residual tile
struct BiasTile {
  f32x4 v[4];
};
__device__ __forceinline__ BiasTile tile_bias_load(const float* __restrict__ bias, const int col0, const int lane) {
  BiasTile b;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    b.v[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) b.v[g] = *reinterpret_cast<const f32x4*>(bias + col0 + 8 * g + 4 * (lane >> 5));
  }
  return b;
}

// acc (32 channels col0.. of 32 tokens, lane = token) = (acc + bias tile) (+ / *) residual tile
template <int RES>  // 0 none, 1 add, 2 multiply
__device__ __forceinline__ void tile_bias_res(f32x16& acc, const BiasTile& bt, const ResTile& rt,
                                              float* __restrict__ wt, const int lane) {
  const int cr = lane >> 3, cp = (lane & 7) * 4;
  const int r = lane & 31, hf = lane >> 5;
  if (RES != 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(wt + (8 * i + cr) * WT_PITCH + cp) = rt.v[i];
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 bv = bt.v[g];
    if (RES != 0) {
      const f32x4 rv = *reinterpret_cast<const f32x4*>(wt + r * WT_PITCH + 8 * g + 4 * hf);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (RES == 1) acc[4 * g + c] = acc[4 * g + c] + bv[c] + rv[c];
        else acc[4 * g + c] = (acc[4 * g + c] + bv[c]) * rv[c];
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[4 * g + c] += bv[c];
    }
  }
  if (RES != 0) __builtin_amdgcn_wave_barrier();
}

// stores acc (32 channels of 32 tokens) in full 128-byte lines
__device__ __forceinline__ void tile_store(const f32x16& acc, float* __restrict__ out, const long long ldo, const int m0,
                                           const int M, const int col0, float* __restrict__ wt, const int lane,
                                           tce_amax_t& amax) {
  const int cr = lane >> 3, cp = (lane & 7) * 4;
  const int r = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    amax = tce_amax4(amax, o);
    *reinterpret_cast<f32x4*>(wt + r * WT_PITCH + 8 * g + 4 * hf) = o;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + 8 * i + cr;
    const f32x4 o = *reinterpret_cast<const f32x4*>(wt + (8 * i + cr) * WT_PITCH + cp);
    if (row < M) *reinterpret_cast<f32x4*>(out + (long long)row * ldo + col0 + cp) = o;
  }
  __builtin_amdgcn_wave_barrier();
}

// Partial sums exchanged between two workgroups of one launch (the split FFN): agent-scope accesses (sc1: written through to /
// read from the memory side of the per-XCD L2s) instead of fences -- an agent-scope release / acquire writes back and INVALIDATES
// the XCD's whole L2, and with it the weight stream every workgroup of the XCD is reading (measured: the split launch 258 us with
// __threadfence() against 167 us un-split).  Register order: lane l's f32x4 number j of the wave at ((j * 64) + l) * 16 bytes, so an
// instruction moves one contiguous KiB and the reader is the same lane of the same wave of the partner workgroup.
__device__ __forceinline__ void partial_store(const f32x16& acc, float* __restrict__ base, const int t, const int lane) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    float* const ptr = base + ((t * 4 + g) * 64 + lane) * 4;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(o) : "memory");
  }
}
struct PartTile {
  f32x4 v[4];
};
// Four tiles (16 loads, one round trip) per asm statement, the wait INSIDE it: between an asm load and a separate wait the compiler
// may move the "loaded" registers (copies into AGPRs under this kernel's register pressure) before the data has arrived -- measured:
// finite garbage in some elements.
__device__ __forceinline__ void partial_load4(const float* __restrict__ base, const int t0, const int lane, PartTile& a, PartTile& b,
                                              PartTile& c, PartTile& d) {
  const float* const pa = base + ((t0 * 4) * 64 + lane) * 4;
  const float* const pb = pa + 4 * 64 * 4;
  const float* const pc = pb + 4 * 64 * 4;
  const float* const pd = pc + 4 * 64 * 4;
  asm volatile(
      "global_load_dwordx4 %0, %16, off sc1\n\tglobal_load_dwordx4 %1, %16, off offset:1024 sc1\n\t"
      "global_load_dwordx4 %2, %16, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %16, off offset:3072 sc1\n\t"
      "global_load_dwordx4 %4, %17, off sc1\n\tglobal_load_dwordx4 %5, %17, off offset:1024 sc1\n\t"
      "global_load_dwordx4 %6, %17, off offset:2048 sc1\n\tglobal_load_dwordx4 %7, %17, off offset:3072 sc1\n\t"
      "global_load_dwordx4 %8, %18, off sc1\n\tglobal_load_dwordx4 %9, %18, off offset:1024 sc1\n\t"
      "global_load_dwordx4 %10, %18, off offset:2048 sc1\n\tglobal_load_dwordx4 %11, %18, off offset:3072 sc1\n\t"
      "global_load_dwordx4 %12, %19, off sc1\n\tglobal_load_dwordx4 %13, %19, off offset:1024 sc1\n\t"
      "global_load_dwordx4 %14, %19, off offset:2048 sc1\n\tglobal_load_dwordx4 %15, %19, off offset:3072 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a.v[0]), "=&v"(a.v[1]), "=&v"(a.v[2]), "=&v"(a.v[3]), "=&v"(b.v[0]), "=&v"(b.v[1]), "=&v"(b.v[2]), "=&v"(b.v[3]),
        "=&v"(c.v[0]), "=&v"(c.v[1]), "=&v"(c.v[2]), "=&v"(c.v[3]), "=&v"(d.v[0]), "=&v"(d.v[1]), "=&v"(d.v[2]), "=&v"(d.v[3])
      : "v"(pa), "v"(pb), "v"(pc), "v"(pd)
      : "memory");
}

// LayerNorm over the N = 32 * NT channels a lane pair (l, l^32) holds in acc[NT] (registers 4g..4g+3 of tile t =
// channels 32t + 8g + 4hf + 0..3)
template <int NT>
__device__ __forceinline__ void rows_layernorm(f32x16 (&acc)[NT], const float* __restrict__ gamma,
                                               const float* __restrict__ beta, const float eps, const int hf) {
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += acc[t][i];
  sum += __shfl_xor(sum, 32, 64);
  const float mean = sum * (1.f / (32 * NT));
  float sq = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float d = acc[t][i] - mean;
      sq = fmaf(d, d, sq);
    }
  sq += __shfl_xor(sq, 32, 64);
  const float rstd = rsqrtf(sq * (1.f / (32 * NT)) + eps);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + 32 * t + 8 * g + 4 * hf);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(beta + 32 * t + 8 * g + 4 * hf);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[t][4 * g + c] = (acc[t][4 * g + c] - mean) * rstd * gv[c] + bv[c];
    }
}

struct FfnArgs {
  const float* x;
  const unsigned char* wpk;
  const float* b2;
  const float *g_in, *be_in, *g_out, *be_out;
  float* out;
  long long ldx, ldo;
  // optional: addend to the first product's input only (position map; a2_rows > 0: row index modulo a2_rows), a residual
  // other than x (NULL = x) combined by res_mode (1 add, 2 multiply), batch strides (grid.y) in floats
  const float *a2, *res;
  long long lda2, ldres, sX, sRes, sOut, sW;  // sW: bytes between the weight streams of consecutive batch entries
  int a2_rows, res_mode;
  int M, NI;  // NI = hidden/32 + 1 iterations
  float eps_in, eps_out;
  int* range_flag;  // tce_set_range_flag: set when a hidden or an output value leaves the fp16 range of the split
  int single;       // tce_set_gemm_mode(2): one MFMA per product on nearest-rounded fp16 operands
  int wdiv;         // batch entries (grid.y) sharing one weight stream: stream index = blockIdx.y / wdiv (>= 1)
  // second stage of a CHAIN (ACT2 != 0, round 5): this kernel's result y (after its LayerNorm) is written to `mid` and, still in
  // registers, becomes the input of an FFN  out = LN2(y + W2' act2(W1' y + b1') + b2')  -- cross-attention -> FFN of a
  // VisionLanguageBlock / a FrameTokenLayer as ONE launch.  wpk2: stream packed with tce_ffn_pack_chain_f32 (W1' in the k order
  // of the accumulator registers); mid must not alias x / res / out of stage 1 that other lanes still read (it is only written
  // and re-read by the lane that owns the rows).
  const unsigned char* wpk2;
  const float *b22, *g_out2, *be_out2;
  float* mid;
  long long ldmid, sMid;
  int NI2;
  float eps_out2;
  // SPLIT (round 5): a 128-row block's hidden extent cut once, the two pieces on different workgroups (sp_p blocks -> sp_p + 1
  // workgroups, cut positions sp_k*), partial sums through ws[nblk][2][rows][C], the block's last arriver (cnt, self-resetting)
  // adds them and runs the epilogue -- see ffn_split_plan
  float* ws;
  int* cnt;
  int sp_p, sp_k0, sp_k1, nblk;  // sp_p = 1 or 2
};

// SINGLE (the arithmetic mode, FfnArgs.single at the launch) is a COMPILE-TIME parameter: as a run-time flag it put one uniform
// branch in front of every step's two lo-term MFMAs (and around every hi / lo split), and the basic-block boundaries kept the
// scheduler from overlapping a step's loads, DMA and split with its MFMAs: 192 -> 175 us at 24100 rows, 90 -> 79 us for the
// C = 192 instantiation, the config-2 clip 6.87 -> 6.63 ms (A/B in one call, profiles/r04_single_template.txt).
template <int C, int WAVES, int ACT, bool SINGLE, int ACT2 = 0, bool SPLIT = false>
__global__ void __launch_bounds__(64 * WAVES, (WAVES == 4 && C <= 128) ? 2 : WAVES / 4) ffn_fused_kernel(const FfnArgs p) {
  static_assert(!(SPLIT && ACT2 != 0), "a chain is not split");
  constexpr int KS = C / 16, NT = C / 32;
  // HALF (round 5): the C <= 128 kernels (packed for 8 waves = 256 rows per workgroup) as 128-row workgroups, TWO per CU, reading the
  // same stream.  72000 rows are 282 workgroups of 256 rows = 1.1 rounds of the chip that cost two; as 563 half workgroups the 51 of
  // the second round run one wave per SIMD and finish in half a round (ffn_launch picks per launch).  Its staging tiles alias ring
  // stage 1: they are used before the loop (stage 0 is being filled) and after it (both stages drained), never inside.
  constexpr bool HALF = WAVES == 4 && C <= 128;
  constexpr int PWAVES = HALF ? 8 : WAVES;  // the stream's padding unit
  constexpr int P = (1 + 2 * KS + 4 * NT + PWAVES - 1) / PWAVES * PWAVES;  // pieces per stage, padded so that every wave issues exactly SLOTS DMAs (no branches)
  constexpr int SLOTS = P / WAVES;
  static_assert(!(HALF && (SPLIT || ACT2 != 0)), "the half form is the plain kernel");
  constexpr int STAGE = P * PIECE;
  constexpr int STEPS = KS + 2 * NT;
  static_assert(SLOTS <= STEPS, "more DMA slots than loop steps");
  static_assert(2 * STAGE + (HALF ? 0 : WAVES * WT_BYTES) <= (HALF ? 80 : 160) * 1024, "ring + staging tiles do not fit the LDS");
  static_assert(!HALF || WAVES * WT_BYTES <= STAGE, "the half form's staging tiles alias one ring stage");
  // the ONLY LDS object (base offset 0): two ring stages, then one staging tile per wave
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE + (HALF ? 0 : WAVES * WT_BYTES) + (SPLIT ? 16 : 0)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hf = lane >> 5;

  // The ring is written only by DMA (inline asm), which the compiler cannot see: without a visible store it treats
  // every read of `smem` as undefined and deletes it.  This store never executes (NI >= 2 always).
  if (p.NI < 0) reinterpret_cast<u32x4*>(smem)[tid] = u32x4{0u, 0u, 0u, 0u};

  long long* const stamps = (g_ffn_stamps && blockIdx.x < 1024 && blockIdx.y == 0 && tid == 0) ? g_ffn_stamps + blockIdx.x * 8 : nullptr;
  if (stamps) {
    stamps[0] = (long long)__builtin_amdgcn_s_memtime();
    stamps[4] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  // this wave's pieces are wave, wave + WAVES, ... of every block, and blocks are contiguous: the source pointer
  // simply advances by WAVES pieces per DMA for the whole kernel (the stream ends with one block of padding, so the
  // last iteration's prefetch reads defined bytes that are never consumed)
  const unsigned char* wp = p.wpk + (long long)(blockIdx.y / p.wdiv) * p.sW + (long long)wave * PIECE;
  const unsigned char* const wp0 = wp;
  const unsigned voff = lane * 16;
  const unsigned wbase = wave * PIECE;
  auto dma = [&](int stage, int q) {
    const unsigned char* src = wp;
    if constexpr (ACT2 != 0 || SPLIT) {
      // the chain switches `wp` to the second stream inside run-time loops: tell the compiler it is still wave-uniform (the DMA's
      // base is an SGPR pair)
      const unsigned long long u = (unsigned long long)wp;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
      src = (const unsigned char*)(((unsigned long long)hi << 32) | lo);
      glds16(src, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(wbase + (unsigned)(stage * STAGE + q * WAVES * PIECE))));
      wp += WAVES * PIECE;
      return;
    }
    glds16(src, voff, wbase + (unsigned)(stage * STAGE + q * WAVES * PIECE));
    wp += WAVES * PIECE;
  };
  const float* const xb = p.x + blockIdx.y * p.sX;
  const float* const resb = p.res ? p.res + blockIdx.y * p.sRes : xb;
  const long long ldres = p.res ? p.ldres : p.ldx;
  float* const outb = p.out + blockIdx.y * p.sOut;
  float* const wt = reinterpret_cast<float*>(smem + (HALF ? STAGE : 2 * STAGE) + wave * WT_BYTES);
  tce_amax_t amax = 0;
  constexpr int single = SINGLE;
  // SPLIT: workgroup i of a group of sp_p + 1 runs the head [0, k_i) of the group's block i, then the tail [k_(i-1), NC) of block
  // i - 1 (a piece over chunks [c0, c1) is the stream's iterations c0 .. c1: the first applies W2[c0 - 1] to a zero H, the last
  // computes an H nobody reads)
  constexpr int NPC = SPLIT ? 2 : 1;
  for (int pc = 0; pc < NPC; ++pc) {
  int blk = blockIdx.x, nit = p.NI;
  if constexpr (SPLIT) {
    const int q = p.sp_p + 1, g = blockIdx.x / q, i = blockIdx.x - g * q;
    int c0 = 0;
    if (pc == 0) {
      if (i == p.sp_p) continue;
      blk = g * p.sp_p + i;
      nit = (i == 0 ? p.sp_k0 : p.sp_k1) + 1;
    } else {
      if (i == 0) continue;
      blk = g * p.sp_p + i - 1;
      c0 = i == 1 ? p.sp_k0 : p.sp_k1;
      nit = p.NI - c0;
    }
    if (blk >= p.nblk) continue;
    wp = wp0 + (long long)c0 * (P * PIECE);
  }
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) dma(0, q);

  // ---- x: 32 tokens x C -> fp16 hi/lo B fragments, full-line loads staged through this wave's private tile
  const int m0 = blk * (32 * WAVES) + wave * 32;
  h16x8 xh[KS], xl[KS];
  load_x_frags<C, (ACT >= 3)>(xb, p.ldx, p.a2, p.lda2, p.a2_rows, m0, p.M, wt, lane, p.g_in, p.be_in, p.eps_in, xh, xl, SINGLE);

  f32x16 oacc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[t][i] = 0.f;
  h16x8 hh0, hl0, hh1, hl1;  // H^T of the previous chunk as B fragments (k-steps 0 and 1)
#pragma unroll
  for (int j = 0; j < 8; ++j) hh0[j] = hl0[j] = hh1[j] = hl1[j] = (_Float16)0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // piece index of step `st` (0..STEPS-1) inside a stage: hi piece; the lo piece follows it
  auto piece_of = [](int st) { return 1 + 2 * st; };
  auto body = [&](auto stage_c, auto act_c) {
    constexpr int A_ = decltype(act_c)::value;  // this stage's activation (the second stage of a chain runs ACT2)
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
        if (!single) {
          hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[s], hacc, 0, 0, 0);
          hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[s], hacc, 0, 0, 0);
        }
        hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[s], hacc, 0, 0, 0);
      } else {
        const int j = s - KS, t = j >> 1;
        if (j & 1) {
          if (!single) {
            oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, hl1, oacc[t], 0, 0, 0);
            oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, hh1, oacc[t], 0, 0, 0);
          }
          oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, hh1, oacc[t], 0, 0, 0);
        } else {
          if (!single) {
            oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, hl0, oacc[t], 0, 0, 0);
            oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, hh0, oacc[t], 0, 0, 0);
          }
          oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, hh0, oacc[t], 0, 0, 0);
        }
        if (A_ == 3 && j == 0) {
          // "activation" = softmax over the chunk's 32 rows (the 32 keys of one attention head) per token: 16 of them
          // in this lane's registers, 16 in lane ^ 32.  In place; the pair loop below then only splits.
          float mx = hacc[0];
#pragma unroll
          for (int i = 1; i < 16; ++i) mx = fmaxf(mx, hacc[i]);
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          float sum = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            hacc[i] = __expf(hacc[i] - mx);
            sum += hacc[i];
          }
          sum += __shfl_xor(sum, 32, 64);
          const float inv = 1.0f / sum;
#pragma unroll
          for (int i = 0; i < 16; ++i) hacc[i] *= inv;
        }
        if (A_ == 4 && j == 0) {
          // 8 keys per head, four heads per chunk: head g of the chunk = rows 8g..8g+7 = registers 4g..4g+3 of this lane
          // (rows 8g + 4hf + 0..3) and of lane ^ 32
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float mx = fmaxf(fmaxf(hacc[4 * g], hacc[4 * g + 1]), fmaxf(hacc[4 * g + 2], hacc[4 * g + 3]));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              hacc[4 * g + i] = __expf(hacc[4 * g + i] - mx);
              sum += hacc[4 * g + i];
            }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int i = 0; i < 4; ++i) hacc[4 * g + i] *= inv;
          }
        }
        // activation + hi/lo split of value pairs [8j/(2NT), 8(j+1)/(2NT)) of this chunk, spread over the 2NT steps
#pragma unroll
        for (int q = (8 * j) / (2 * NT); q < (8 * (j + 1)) / (2 * NT); ++q) {
          float v0 = hacc[2 * q], v1 = hacc[2 * q + 1];
          if (A_ == 1) {  // plain v_max: fmaxf() would first quiet a possible sNaN with a second v_max per value
            asm("v_max_f32_e32 %0, 0, %1" : "=v"(v0) : "v"(v0));
            asm("v_max_f32_e32 %0, 0, %1" : "=v"(v1) : "v"(v1));
          }
          if (A_ == 2) {
            v0 = tce_gelu(v0);
            v1 = tce_gelu(v1);
          }
          amax = max(amax, max(tce_absbits(v0), tce_absbits(v1)));
          const fp16x2_t a = single ? fp16x2_t{(__fp16)v0, (__fp16)v1} : __builtin_amdgcn_cvt_pkrtz(v0, v1);
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
  using act1_t = std::integral_constant<int, ACT>;
  if constexpr (ACT2 == 0) {
    int it = 0;
    for (; it + 1 < nit; it += 2) {
      body(std::integral_constant<int, 0>{}, act1_t{});
      body(std::integral_constant<int, 1>{}, act1_t{});
    }
    if (it < nit) body(std::integral_constant<int, 0>{}, act1_t{});
  } else {
    // ---- chain: stage 1, its epilogue into `mid` and into the operand registers, stage 2.  The ring parity is a run-time value
    // here (one uniform branch per iteration of 96 MFMAs).
    int par = 0;
    auto run = [&](const int n, auto act_c) {
      for (int i = 0; i < n; ++i) {
        if (par == 0) body(std::integral_constant<int, 0>{}, act_c);
        else body(std::integral_constant<int, 1>{}, act_c);
        par ^= 1;
      }
    };
    run(p.NI - 1, act1_t{});
    wp = p.wpk2 + (long long)wave * PIECE;  // the last iteration of stage 1 prefetches block 0 of stage 2's stream
    run(1, act1_t{});
    {
      float* const midb = p.mid + blockIdx.y * p.sMid;
      ResTile rt[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) rt[t] = tile_res_load(resb, ldres, m0, p.M, 32 * t, lane);
      if (p.res_mode == 2) {
#pragma unroll
        for (int t = 0; t < NT; ++t) tile_bias_res<2>(oacc[t], tile_bias_load(p.b2, 32 * t, lane), rt[t], wt, lane);
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) tile_bias_res<1>(oacc[t], tile_bias_load(p.b2, 32 * t, lane), rt[t], wt, lane);
      }
      if (p.g_out) rows_layernorm<NT>(oacc, p.g_out, p.be_out, p.eps_out, hf);
#pragma unroll
      for (int t = 0; t < NT; ++t) tile_store(oacc[t], midb, p.ldmid, m0, p.M, 32 * t, wt, lane, amax);
      // y (accumulator layout: lane = token, registers 4g..4g+3 of tile t = channels 32t + 8g + 4hf + 0..3) IS a B operand in
      // the k order  16s + 8(j>>2) + 4hf + (j&3)  -- the order stage 2's W1 fragments are packed in (tce_ffn_pack_chain_f32)
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
        const int t = s2 >> 1, g0 = 2 * (s2 & 1);
        const float f[8] = {oacc[t][4 * g0], oacc[t][4 * g0 + 1], oacc[t][4 * g0 + 2], oacc[t][4 * g0 + 3],
                            oacc[t][4 * g0 + 4], oacc[t][4 * g0 + 5], oacc[t][4 * g0 + 6], oacc[t][4 * g0 + 7]};
        const HL sp = split8(f, single);
        xh[s2] = sp.hi;
        xl[s2] = sp.lo;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[t][i] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) hh0[j] = hl0[j] = hh1[j] = hl1[j] = (_Float16)0.f;
    }
    run(p.NI2, std::integral_constant<int, ACT2>{});
  }
  if (FFN_ABL & 3) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (stamps) stamps[2] = (long long)__builtin_amdgcn_s_memtime();

  if constexpr (SPLIT) {
    // this piece's partial sums -> ws[blk][role]; the block's second arriver adds the other piece's (a + b: the same bits
    // whichever piece arrives last) and goes on to the epilogue, the first arriver is done with the block
    const int role = pc;
    const long long bi = (long long)blockIdx.y * p.nblk + blk;
    float* const mine = p.ws + (bi * 2 + role) * (32 * WAVES * C);
    const float* const other = p.ws + (bi * 2 + (role ^ 1)) * (32 * WAVES * C);
    float* const mine_w = mine + wave * (32 * C);
    const float* const other_w = other + wave * (32 * C);
#pragma unroll
    for (int t = 0; t < NT; ++t) partial_store(oacc[t], mine_w, t, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // written through: visible to the partner before the counter moves
    __syncthreads();
    int* const flag = reinterpret_cast<int*>(smem + 2 * STAGE + WAVES * WT_BYTES);
    if (tid == 0) *flag = __hip_atomic_fetch_add(p.cnt + bi, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int arrived = __builtin_amdgcn_readfirstlane(*flag);
    __syncthreads();  // (the flag word is rewritten by the next piece)
    if (arrived == 0) continue;
    if (tid == 0) __hip_atomic_store(p.cnt + bi, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    static_assert(NT % 4 == 0, "the split kernel exchanges four tiles at a time");
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += 4) {
      PartTile ot[4];
      partial_load4(other_w, t0, lane, ot[0], ot[1], ot[2], ot[3]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int c = 0; c < 4; ++c) oacc[t0 + t][4 * g + c] += ot[t].v[g][c];
    }
  }
  // ---- epilogue: + b2 + residual, optional LayerNorm, stores -- all global traffic in full lines through the wave's
  // staging tile
  if constexpr (ACT2 != 0) {  // stage 2 of a chain: residual = this lane's own rows of `mid`, additive; stage 2's bias / LayerNorm
    const float* const midb = p.mid + blockIdx.y * p.sMid;
    ResTile rt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) rt[t] = tile_res_load(midb, p.ldmid, m0, p.M, 32 * t, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t) tile_bias_res<1>(oacc[t], tile_bias_load(p.b22, 32 * t, lane), rt[t], wt, lane);
    if (p.g_out2) rows_layernorm<NT>(oacc, p.g_out2, p.be_out2, p.eps_out2, hf);
  } else {
    ResTile rt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) rt[t] = tile_res_load(resb, ldres, m0, p.M, 32 * t, lane);
    if (p.res_mode == 2) {
#pragma unroll
      for (int t = 0; t < NT; ++t) tile_bias_res<2>(oacc[t], tile_bias_load(p.b2, 32 * t, lane), rt[t], wt, lane);
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) tile_bias_res<1>(oacc[t], tile_bias_load(p.b2, 32 * t, lane), rt[t], wt, lane);
    }
    if (p.g_out) rows_layernorm<NT>(oacc, p.g_out, p.be_out, p.eps_out, hf);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) tile_store(oacc[t], outb, p.ldo, m0, p.M, 32 * t, wt, lane, amax);
  }  // pieces
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
                                                       const long long units, const int single, const int permk = 0) {
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  if (u >= units) return;
  W1 += (long long)blockIdx.y * Hd * C;  // batch entries: contiguous [Hd,C], [Hd], [C,Hd] blocks, contiguous streams
  if (b1) b1 += (long long)blockIdx.y * Hd;
  W2 += (long long)blockIdx.y * C * Hd;
  out += (long long)blockIdx.y * units * 16;
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
      for (int j = 0; j < 8; ++j) {
        // permk (second stage of a chain): the x operand is the previous stage's accumulator, whose registers enumerate a
        // k-step's 16 channels as 8(j>>2) + 4hf + (j&3) -- W1 follows that order (like W2 does for the hidden units below)
        const int kk = permk ? 16 * s + 8 * (j >> 2) + 4 * hf + (j & 3) : 16 * s + 8 * hf + j;
        v[j] = it < NC ? W1[(long long)(32 * it + r) * C + kk] : 0.f;
      }
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
    const HL f = split8(v, single);
    o = __builtin_bit_cast(u32x4, plane ? f.lo : f.hi);
  }
  reinterpret_cast<u32x4*>(out)[u] = o;
}

// ---------------------------------------------------------------------------------------------------------------
// Token-stationary linear layer ("rowlin"):  out[M,N] = LN_out?( epi( LN_in?(x + a2) W^T + bias ) )   K = 96..256
// Same machinery as the fused FFN: x (32 tokens per wave, K/2 registers) is loaded once in full lines and stays in
// registers as fp16 hi/lo B fragments; W [N,K] is pre-split / pre-ordered into 1 KiB A fragments, one block of
// 2*K/16 pieces per 32 output channels, streamed L2 -> LDS by DMA through a two-stage ring; each 32-channel
// accumulator tile is finished (bias, activation, residual add / multiply) and stored in full lines through the
// wave's staging tile while the next tile's weights are in flight.  ROW mode (N = 256) keeps all 8 tiles in registers
// and applies a LayerNorm over the row before storing: out_proj + residual + LayerNorm of the post-norm transformer
// blocks in one launch.  Replaces the K <= 256 nn.Linear / 1x1 conv call sites with many rows
// (tce_deformable_transformer.py:439-489,535-548; ops/modules/ms_deform_attn.py:94-101,115; segmentation.py:187,330-372;
// swin_transformer.py:133,151; tce_rvos.py:260).
// ---------------------------------------------------------------------------------------------------------------
struct LinArgs {
  const float *x, *a2;
  const unsigned char* wpk;
  const float *bias, *res;
  float* out;
  const float *g_in, *be_in, *g_out, *be_out;
  long long ldx, lda2, ldres, ldo;
  long long sX, sA2, sRes, sOut;  // batch strides in floats (grid.y)
  int M, N, a2_rows, act, res_mode;
  float eps_in, eps_out;
  int* range_flag;
  int single;
};

// K = 384 (round 4: Swin-T's third stage, 4600 rows at config 2: norm1 -> qkv, proj + residual, norm2 -> fc1 + GELU): the x
// fragments take 192 registers -> one workgroup per CU (90 KB of LDS); with the output tiles split over gridDim.z the 36 row
// blocks become ~290 workgroups of 4-6 tiles each, against the tiled GEMM's 12 short K slices per tile (32 us per launch there).
// K = 96 (Swin-T stage 0) is compiled for THREE workgroups per CU (168 registers, 43 KB of LDS): at config 2 the stage has
// 72000 rows = 563 row blocks, one more than two per CU hold (512) -- a second, nearly empty round.  47.7 -> 38.5 us for the
// norm1 -> qkv launch (profiles/r03_rowlin_occupancy.txt).
template <int K, bool ROW, bool SINGLE>  // SINGLE: compile-time arithmetic mode, see ffn_fused_kernel
__global__ void __launch_bounds__(256, (ROW || K > 256) ? 1 : (K <= 96 ? 3 : 2)) rowlin_kernel(const LinArgs p) {
  // The ring holds HALF blocks (the first / second K/32 k-steps of a 32-channel tile, hi and lo pieces interleaved,
  // padded to a multiple of 4 pieces so that every wave issues the same number of DMAs): three half-stages, the DMA of
  // half h+2 is issued while half h is multiplied.  3 x 16 KiB + staging tiles = 66 KiB at K = 256: two workgroups
  // per CU in streaming mode (<= 256 registers), so one workgroup's tile epilogue hides under the other's MFMAs.
  constexpr int WAVES = 4, KS = K / 16, KH = KS / 2;
  constexpr int HP = (2 * KH + WAVES - 1) / WAVES * WAVES;  // pieces per half block
  constexpr int SLOTS = HP / WAVES;
  constexpr int HSTAGE = HP * PIECE;
  constexpr int NTR = 8;  // ROW mode: N = 256
  static_assert(KS % 2 == 0 && SLOTS <= KH, "piece count");
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * HSTAGE + WAVES * WT_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hf = lane >> 5;
  const int bz = blockIdx.y;
  const int m0 = blockIdx.x * (32 * WAVES) + wave * 32;
  if (p.M < 0) reinterpret_cast<u32x4*>(smem)[tid] = u32x4{0u, 0u, 0u, 0u};  // see ffn_fused_kernel

  const float* const x = p.x + bz * p.sX;
  const float* const a2 = p.a2 ? p.a2 + bz * p.sA2 : nullptr;
  const float* const res = p.res ? p.res + bz * p.sRes : nullptr;
  float* const out = p.out + bz * p.sOut;
  // streaming mode: the 32-channel tiles may be split over gridDim.z workgroups (each re-reads x: a few MB against a
  // much shorter serial chain of tiles per workgroup, which is what bounds launches with few rows)
  const int ntiles = p.N / 32;
  const int t_begin = ROW ? 0 : (int)((long long)blockIdx.z * ntiles / gridDim.z);
  const int t_end = ROW ? ntiles : (int)((long long)(blockIdx.z + 1) * ntiles / gridDim.z);

  long long* const stamps = (g_ffn_stamps && blockIdx.x < 1024 && bz == 0 && blockIdx.z == 0 && tid == 0) ? g_ffn_stamps + blockIdx.x * 8 : nullptr;
  if (stamps) {
    stamps[0] = (long long)__builtin_amdgcn_s_memtime();
    stamps[4] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  const unsigned char* wp = p.wpk + ((long long)2 * t_begin * HP + wave) * PIECE;
  const unsigned voff = lane * 16;
  const unsigned wbase = wave * PIECE;
  int dstage = 0;  // ring slot the next half block goes to
  auto dma_half = [&](int q) { glds16(wp + (long long)q * WAVES * PIECE, voff, wbase + (unsigned)(dstage * HSTAGE + q * WAVES * PIECE)); };
  auto dma_next = [&]() {
    wp += HP * PIECE;
    dstage = dstage == 2 ? 0 : dstage + 1;
  };
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) dma_half(q);
    dma_next();
  }

  float* const wt = reinterpret_cast<float*>(smem + 3 * HSTAGE + wave * WT_BYTES);
  h16x8 xh[KS], xl[KS];
  load_x_frags<K>(x, p.ldx, a2, p.lda2, p.a2_rows, m0, p.M, wt, lane, p.g_in, p.be_in, p.eps_in, xh, xl, SINGLE);
  constexpr int single = SINGLE;
  tce_amax_t amax = 0;
  if (stamps) stamps[6] = (long long)__builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (stamps) stamps[1] = (long long)__builtin_amdgcn_s_memtime();

  int cstage = 0;  // ring slot of the half block being multiplied
  // one half of a 32-channel tile: acc += W[tile, k-steps of this half] x^T; issues the DMA of the half after next.
  // Fragments are fetched TWO k-steps ahead of their MFMAs (LDS latency under load is longer than one step's three
  // MFMAs); sched_barrier pins each step, otherwise the compiler sinks the reads behind the MFMAs they should overlap.
  auto half_mma = [&](auto half_c, f32x16& acc) {
    constexpr int S0 = decltype(half_c)::value * KH;
    const unsigned char* const st = smem + cstage * HSTAGE + lane * 16;
    h16x8 fh[3], fl[3];
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (q < KH) {
        fh[q] = *reinterpret_cast<const h16x8*>(st + (2 * q) * PIECE);
        fl[q] = *reinterpret_cast<const h16x8*>(st + (2 * q + 1) * PIECE);
      }
#pragma unroll
    for (int s = 0; s < KH; ++s) {
      if (s + 2 < KH) {
        fh[(s + 2) % 3] = *reinterpret_cast<const h16x8*>(st + (2 * s + 4) * PIECE);
        fl[(s + 2) % 3] = *reinterpret_cast<const h16x8*>(st + (2 * s + 5) * PIECE);
      }
      if (s < SLOTS) dma_half(s);
      __builtin_amdgcn_sched_barrier(0);
      if (!single) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[s % 3], xl[S0 + s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[s % 3], xh[S0 + s], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[s % 3], xh[S0 + s], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    dma_next();
    cstage = cstage == 2 ? 0 : cstage + 1;
  };
  auto act_tile = [&](f32x16& acc) {
    if (p.act == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fmaxf(acc[i], 0.f);
    } else if (p.act == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = tce_gelu(acc[i]);
    }
  };
  // bias, activation, residual (activation sits between bias and residual, as in tce_gemm_f32)
  auto finish_tile = [&](f32x16& acc, const BiasTile& bt, const ResTile& rt) {
    const ResTile none = {};
    const BiasTile nob = {};
    if (p.res_mode == 0 || p.act != 0) {
      tile_bias_res<0>(acc, bt, none, wt, lane);
      act_tile(acc);
      if (p.res_mode == 1) tile_bias_res<1>(acc, nob, rt, wt, lane);
      else if (p.res_mode == 2) tile_bias_res<2>(acc, nob, rt, wt, lane);
    } else if (p.res_mode == 1) {
      tile_bias_res<1>(acc, bt, rt, wt, lane);
    } else {
      tile_bias_res<2>(acc, bt, rt, wt, lane);
    }
  };
  // the half just multiplied may be overwritten and the next one must have landed: only this wave's DMAs of the half
  // after next (SLOTS of them) may stay in flight
  auto tile_mma = [&](f32x16& acc, const bool stores_follow) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    half_mma(std::integral_constant<int, 0>{}, acc);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS) : "memory");
    __syncthreads();
    half_mma(std::integral_constant<int, 1>{}, acc);
    (void)stores_follow;
  };

  if (ROW) {
    f32x16 oacc[NTR];
#pragma unroll
    for (int t = 0; t < NTR; ++t) {
      tile_mma(oacc[t], false);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS) : "memory");
      __syncthreads();
    }
    {
      ResTile rt[NTR];
      if (p.res_mode != 0) {
#pragma unroll
        for (int t = 0; t < NTR; ++t) rt[t] = tile_res_load(res, p.ldres, m0, p.M, 32 * t, lane);
      }
#pragma unroll
      for (int t = 0; t < NTR; ++t) finish_tile(oacc[t], tile_bias_load(p.bias, 32 * t, lane), rt[t]);
    }
    if (p.g_out) rows_layernorm<NTR>(oacc, p.g_out, p.be_out, p.eps_out, hf);
#pragma unroll
    for (int t = 0; t < NTR; ++t) tile_store(oacc[t], out, p.ldo, m0, p.M, 32 * t, wt, lane, amax);
  } else {
    for (int t = t_begin; t < t_end; ++t) {
      ResTile rt = {};
      if (p.res_mode != 0) rt = tile_res_load(res, p.ldres, m0, p.M, 32 * t, lane);  // both land under the MFMAs
      const BiasTile bt = tile_bias_load(p.bias, 32 * t, lane);
      f32x16 acc;
      tile_mma(acc, true);
      // Only the DMAs of the half after next are younger than what must have landed, and DMAs retire in issue order
      // among themselves -- so this count is exact.  The wait sits BEFORE the tile's stores on purpose: loads and
      // stores do not retire in order relative to each other, and with the stores in front of it a count of
      // SLOTS + 4 could be satisfied by early store acknowledgements while a needed DMA was still in flight.
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS) : "memory");
      __syncthreads();
      finish_tile(acc, bt, rt);
      tile_store(acc, out, p.ldo, m0, p.M, 32 * t, wt, lane, amax);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may outlive the workgroup's LDS allocation
  tce_range_report(p.range_flag, amax);
  if (stamps) {
    stamps[2] = (long long)__builtin_amdgcn_s_memtime();
    stamps[3] = (long long)__builtin_amdgcn_s_memtime();
    stamps[5] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}

// W [N, K] (row pitch ldw) -> per 32 output channels two HALF blocks of HP pieces (HP = 2 * K/32 rounded up to a
// multiple of 4): piece 2j (+1) of half h = hi (lo) fragment of k-step s = h*K/32 + j: lane (r, hf) holds
// W[32t + r][16s + 8hf + 0..7]; padding pieces are zero.  Two trailing half blocks of zeros (the prefetch runs two
// halves ahead).
__global__ void __launch_bounds__(256) rowlin_pack_kernel(const float* __restrict__ W, unsigned char* __restrict__ out,
                                                          const int N, const int K, const long long ldw,
                                                          const long long units, const int single) {
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  if (u >= units) return;
  const int KH = K / 32, HP = (2 * KH + 3) / 4 * 4;
  const int lane = (int)(u & 63);
  const long long pg = u >> 6;
  const int piece = (int)(pg % HP);
  const long long hb = pg / HP;  // half-block index = 2 * tile + half
  const int t = (int)(hb >> 1), h = (int)(hb & 1);
  const int r = lane & 31, hf = lane >> 5, s = h * KH + (piece >> 1);
  u32x4 o = {0u, 0u, 0u, 0u};
  const int n = 32 * t + r;
  if (piece < 2 * KH && n < N) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = W[(long long)n * ldw + 16 * s + 8 * hf + j];
    const HL f = split8(v, single);
    o = __builtin_bit_cast(u32x4, (piece & 1) ? f.lo : f.hi);
  }
  reinterpret_cast<u32x4*>(out)[u] = o;
}

inline bool rowlin_shape_ok(int N, int K) {
  return (K == 96 || K == 128 || K == 192 || K == 256 || K == 384 || K == 512) && N > 0 && N % 32 == 0;
}
inline long long rowlin_units(int N, int K) {
  const int HP = (2 * (K / 32) + 3) / 4 * 4;
  return (long long)(2 * (N / 32) + 2) * HP * 64;
}

template <int K>
void rowlin_launch(const LinArgs& a, int batch, bool row, hipStream_t s) {
  // aim at >= ~3 workgroups per CU-pair slot: split the output tiles when the row blocks alone leave the chip idle
  const int rb = tce_cdiv(a.M, 128) * batch, ntiles = a.N / 32;
  int nz = 1;
  if (!row && K > 256) {
    // one workgroup per CU (x alone takes 192 registers): exactly ONE round of <= 256 workgroups, at least two tiles each
    // (36 row blocks x 8 splits = 288 workgroups ran a second, nearly empty round: slower than the tiled GEMM)
    nz = 256 / (rb > 0 ? rb : 1);
    if (nz > ntiles / 2) nz = ntiles / 2;
    if (nz < 1) nz = 1;
  } else if (!row) {
    while (nz < ntiles && rb * nz < 160 && ntiles / (nz * 2) >= 2) nz *= 2;  // only launches with few row blocks
  }
  const dim3 grid(tce_cdiv(a.M, 128), batch, nz), block(256);
  if constexpr (K <= 384) {  // (row mode keeps 8 accumulator tiles beside x: no room at K = 512; the entry point rejects it)
    if (row) {
      if (a.single) hipLaunchKernelGGL((rowlin_kernel<K, true, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((rowlin_kernel<K, true, false>), grid, block, 0, s, a);
      return;
    }
  }
  if (a.single) hipLaunchKernelGGL((rowlin_kernel<K, false, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((rowlin_kernel<K, false, false>), grid, block, 0, s, a);
}

inline bool ffn_shape_ok(int C, int Hd) { return (C == 96 || C == 128 || C == 192 || C == 256) && Hd > 0 && Hd % 32 == 0; }
inline int ffn_waves(int C) { return C <= 128 ? 8 : 4; }
inline int ffn_pieces(int C) {
  const int w = ffn_waves(C);
  return (1 + 2 * (C / 16) + 4 * (C / 32) + w - 1) / w * w;
}
inline long long ffn_units(int C, int Hd) { return (long long)(Hd / 32 + 2) * ffn_pieces(C) * 64; }  // + 1 block of padding

// Hidden-extent split of the fused FFN (round 5).  A workgroup owns 128 rows and walks the whole hidden extent: 24100 rows are 189
// workgroups -- 0.74 rounds of the chip's 256 CUs, and a round costs the same whether it is full or not.  With `p` blocks cut once
// each and handed to p + 1 workgroups (head of block i, tail of block i - 1) a round is shorter by ~p / (p + 1): the plan that
// minimises rounds x (iterations of the busiest workgroup + prologue / epilogue cost, in iterations) wins, and must win by 25 % ON
// THAT MODEL: the kernel's throughput is nearly flat in the number of busy CUs (141 / 189 / 563 workgroups: 1.02 / 1.12 / 1.22
// workgroups per us -- the power-bound finding of DESIGN 3.2), so the measured gains are a fraction of the model's: 18000 rows
// (2 -> 3) 138 -> 121 us, 40800 rows (2 -> 3) 288 -> 251 us and config 3's clip 9.82 -> 9.60 ms, but 24100 rows (3 -> 4) only
// 169 -> 163 us and config 2's clip 6.08 -> 6.23 ms with it (the 67 CUs the un-split launch leaves idle are where the clip's
// parallel branches run).  Hence the margin: the 2 -> 3 and 1 -> 2 plans pass it, 3 -> 4 does not.
struct FfnSplitPlan {
  int p = 0, k[2] = {0, 0}, nblk = 0;
  long long ws_floats = 0;
};
inline FfnSplitPlan ffn_split_plan(int M, int C, int Hd, int batch) {
  FfnSplitPlan best;
  const int B = tce_cdiv(M, 32 * ffn_waves(C)), NC = Hd / 32, CUS = 256;
  if (C != 256 || NC < 24 || batch < 1) return best;
  const int OVP = 3, OVF = 4;  // per piece: x load + split + partial store; per finished block: partial read + epilogue
  auto rounds = [&](long long wgs) { return (double)((wgs + CUS - 1) / CUS); };
  double best_cost = 0.75 * rounds((long long)B * batch) * (NC + 1 + OVP + OVF);
  for (int p = 1; p <= 2; ++p) {  // (3 -> 4 was built and measured: see above)
    const int wd = p == 1 ? 0 : (p * NC - 2 * (1 + OVP)) / (p + 1), ws_ = p == 1 ? (NC + 1) / 2 : wd + 1 + OVP;
    int k[2] = {ws_, 0};
    bool ok = ws_ >= 1 && ws_ <= NC - 1;
    for (int i = 1; i < p && ok; ++i) {
      k[i] = wd - (NC - k[i - 1]);
      ok = k[i] >= 1 && k[i] <= NC - 1;
    }
    if (!ok) continue;
    const int busiest = ws_ > NC - k[p - 1] ? ws_ : NC - k[p - 1];
    const double cost = rounds((long long)tce_cdiv(B, p) * (p + 1) * batch) * (busiest + 1 + OVP + OVF);
    if (cost < best_cost) {
      best_cost = cost;
      best.p = p;
      for (int i = 0; i < 2; ++i) best.k[i] = k[i];
    }
  }
  if (best.p) {
    best.nblk = B;
    best.ws_floats = (long long)batch * B * 2 * (32 * ffn_waves(C)) * C;
  }
  return best;
}

static int g_ffn_half = 0;  // tce_debug_ffn_set_half: 0 automatic, 1 always (C <= 128), -1 never

template <int C, int WAVES>
void ffn_launch(const FfnArgs& a, int act, hipStream_t s, int batch = 1) {
  if constexpr (C == 256) {
    if (a.sp_p > 0 && act == 1) {
      const dim3 grid(tce_cdiv(a.nblk, a.sp_p) * (a.sp_p + 1), batch), block(64 * WAVES);
      if (a.single) hipLaunchKernelGGL((ffn_fused_kernel<C, WAVES, 1, true, 0, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((ffn_fused_kernel<C, WAVES, 1, false, 0, true>), grid, block, 0, s, a);
      return;
    }
  }
  if constexpr (C <= 128 && WAVES == 8) {
    // 256-row workgroups or 128-row ones, two per CU (HALF, same stream): by the rounds each form needs -- a last round of at most
    // 256 half workgroups runs one wave per SIMD and costs half a round.  Ties go to the half form (tools/ffn_bench.py: 72000 x 96
    // 92.7 -> 67.4 us, where it saves half a round; 122880 x 96 93.9 -> 88.8, 256800 x 128 279 -> 265, 36000 x 96 42.1 -> 42.3 us
    // at equal round counts)
    const long long b8 = (long long)tce_cdiv(a.M, 256) * batch, b4 = (long long)tce_cdiv(a.M, 128) * batch;
    const double c8 = (double)((b8 + 255) / 256);
    const long long rem = b4 % 512;
    const double c4 = (double)(b4 / 512) + (rem == 0 ? 0.0 : rem <= 256 ? 0.5 : 1.0);
    if (act == 2 && g_ffn_half >= 0 && (g_ffn_half > 0 || c4 <= c8)) {
      const dim3 grid(tce_cdiv(a.M, 128), batch), block(256);
      if (a.single) hipLaunchKernelGGL((ffn_fused_kernel<C, 4, 2, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((ffn_fused_kernel<C, 4, 2, false>), grid, block, 0, s, a);
      return;
    }
  }
  const dim3 grid(tce_cdiv(a.M, 32 * WAVES), batch), block(64 * WAVES);
  if (a.single) {
    if (act == 1) hipLaunchKernelGGL((ffn_fused_kernel<C, WAVES, 1, true>), grid, block, 0, s, a);
    else if (act == 2) hipLaunchKernelGGL((ffn_fused_kernel<C, WAVES, 2, true>), grid, block, 0, s, a);
  } else {
    if (act == 1) hipLaunchKernelGGL((ffn_fused_kernel<C, WAVES, 1, false>), grid, block, 0, s, a);
    else if (act == 2) hipLaunchKernelGGL((ffn_fused_kernel<C, WAVES, 2, false>), grid, block, 0, s, a);
  }
}

// Text cross-attention as an FFN-shaped chain (see tce_xattn_prepare_f32 in the header): folds the per-clip key / value
// rows of every head into the two weight matrices.  One thread per output element, exact fp32 FMAs.
__global__ void __launch_bounds__(256) xattn_prepare_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                            const float* __restrict__ wqT, const float* __restrict__ wo,
                                                            float* __restrict__ W1, float* __restrict__ b1,
                                                            float* __restrict__ W2, const int L, const int G) {
  // G = key slots per head (32 or 8), hidden = 8 * G; blockIdx.y = batch entry (its own keys / values and outputs)
  constexpr int C = 256, HD = 32;
  const int Hd = 8 * G;
  k += (long long)blockIdx.y * L * C;
  v += (long long)blockIdx.y * L * C;
  W1 += (long long)blockIdx.y * Hd * C;
  b1 += (long long)blockIdx.y * Hd;
  W2 += (long long)blockIdx.y * C * Hd;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < Hd * C) {  // W1[h*G + j][i] = sum_c k[j][h*32 + c] * wqT[i][h*32 + c]
    const int r = idx / C, i = idx - r * C, h = r / G, j = r - h * G;
    float a = 0.f;
    if (j < L) {
      const f32x4* kp = reinterpret_cast<const f32x4*>(k + j * C + h * HD);
      const f32x4* wp = reinterpret_cast<const f32x4*>(wqT + (long long)i * C + h * HD);
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const f32x4 kv = kp[c4], wv = wp[c4];
        a = fmaf(kv[0], wv[0], a); a = fmaf(kv[1], wv[1], a); a = fmaf(kv[2], wv[2], a); a = fmaf(kv[3], wv[3], a);
      }
    }
    W1[idx] = a;
  } else if (idx < 2 * Hd * C) {  // W2[n][h*G + j] = sum_c wo[n][h*32 + c] * v[j][h*32 + c]
    const int e = idx - Hd * C, n = e / Hd, r = e - n * Hd, h = r / G, j = r - h * G;
    float a = 0.f;
    if (j < L) {
      const f32x4* vp = reinterpret_cast<const f32x4*>(v + j * C + h * HD);
      const f32x4* wp = reinterpret_cast<const f32x4*>(wo + (long long)n * C + h * HD);
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const f32x4 vv = vp[c4], wv = wp[c4];
        a = fmaf(vv[0], wv[0], a); a = fmaf(vv[1], wv[1], a); a = fmaf(vv[2], wv[2], a); a = fmaf(vv[3], wv[3], a);
      }
    }
    W2[e] = a;
  } else if (idx < 2 * Hd * C + Hd) {  // b1[h*G + j] = sum_c k[j][h*32 + c] * wqT[256][h*32 + c]; padded keys: -inf
    const int r = idx - 2 * Hd * C, h = r / G, j = r - h * G;
    float a = -1.0e30f;
    if (j < L) {
      a = 0.f;
      for (int c = 0; c < HD; ++c) a = fmaf(k[j * C + h * HD + c], wqT[(long long)C * C + h * HD + c], a);
    }
    b1[r] = a;
  }
}

// The fold (xattn_prepare_kernel) and the packing (ffn_pack_kernel) of a short-key cross-attention site in ONE launch: a thread
// produces one 16-byte unit of the weight stream and computes its eight folded values itself (32-term dot products of the
// site's projected keys / values with the static projection weights, the same fmaf order as xattn_prepare_kernel: the stream
// is bit-identical to prepare -> pack).  Nine launches per clip fewer (five text sites, four frame-token layers), two of them
// per encoder layer on the critical frame-token path.
__global__ void __launch_bounds__(256) xattn_pack_fused_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                               const float* __restrict__ wqT, const float* __restrict__ wo,
                                                               unsigned char* __restrict__ out, const int L, const int G,
                                                               const int P, const int SL, const long long threads,
                                                               const long long units, const int single) {
  // a thread owns one SLOT of a stage: slot 0 = the bias piece, slot sl >= 1 = the (hi, lo) piece pair 2 sl - 1, 2 sl -- the
  // eight folded values are computed once and split into both planes (a thread per piece computed every value twice)
  constexpr int C = 256, HD = 32, KS = C / 16, NT = C / 32;
  const long long tix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (tix >= threads) return;
  const int Hd = 8 * G, NC = Hd / 32;
  k += (long long)blockIdx.y * L * C;
  v += (long long)blockIdx.y * L * C;
  out += (long long)blockIdx.y * units * 16;
  const int lane = (int)(tix & 63);
  const long long sg = tix >> 6;
  const int sl = (int)(sg % SL), it = (int)(sg / SL);
  const int r = lane & 31, hf = lane >> 5;
  u32x4* const stage_out = reinterpret_cast<u32x4*>(out) + (long long)it * P * 64;
  auto dot32 = [](const float* a, const float* b) {
    const f32x4* ap = reinterpret_cast<const f32x4*>(a);
    const f32x4* bp = reinterpret_cast<const f32x4*>(b);
    float acc = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
      const f32x4 x = ap[c4], y = bp[c4];
      acc = fmaf(x[0], y[0], acc); acc = fmaf(x[1], y[1], acc); acc = fmaf(x[2], y[2], acc); acc = fmaf(x[3], y[3], acc);
    }
    return acc;
  };
  if (sl == 0) {  // b1 chunk `it`: [hf][16] floats in accumulator order; -1e30 for key slots >= L
    u32x4 o = {0u, 0u, 0u, 0u};
    if (lane < 8 && it < NC) {
      f32x4 val;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int idx = 4 * lane + c, h2 = idx >> 4, i = idx & 15;
        const int row = 32 * it + (i & 3) + 8 * (i >> 2) + 4 * h2, h = row / G, j = row - h * G;
        float a = -1.0e30f;
        if (j < L) {
          a = 0.f;
          for (int c2 = 0; c2 < HD; ++c2) a = fmaf(k[j * C + h * HD + c2], wqT[(long long)C * C + h * HD + c2], a);
        }
        val[c] = a;
      }
      o = __builtin_bit_cast(u32x4, val);
    }
    stage_out[lane] = o;
    return;
  }
  const int p_hi = 2 * sl - 1, p_lo = 2 * sl;  // pieces of this slot
  if (p_hi >= P) return;
  u32x4 oh = {0u, 0u, 0u, 0u}, ol = oh;
  if (p_hi < 1 + 2 * KS + 4 * NT) {
    float val[8];
    if (p_hi <= 2 * KS) {  // W1[32 it + r][16 s + 8 hf + e] = sum_c k[key][h*32 + c] * wqT[column][h*32 + c]
      const int s = (p_hi - 1) >> 1;
      const int row = 32 * it + r, h = row / G, j = row - h * G;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        val[e] = (it < NC && j < L) ? dot32(k + j * C + h * HD, wqT + (long long)(16 * s + 8 * hf + e) * C + h * HD) : 0.f;
    } else {  // W2[32 t + r][col] = sum_c wo[n][h*32 + c] * v[key][h*32 + c], col in the accumulator's k order
      const int idx = (p_hi - 1 - 2 * KS) >> 1, t = idx >> 1, s2 = idx & 1, chunk = it - 1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = 32 * chunk + 16 * s2 + 8 * (e >> 2) + 4 * hf + (e & 3);
        const int h = col / G, j = col - h * G;
        val[e] = (chunk >= 0 && chunk < NC && j < L) ? dot32(wo + (long long)(32 * t + r) * C + h * HD, v + j * C + h * HD) : 0.f;
      }
    }
    const HL f = split8(val, single);
    oh = __builtin_bit_cast(u32x4, f.hi);
    ol = __builtin_bit_cast(u32x4, f.lo);
  }
  stage_out[p_hi * 64 + lane] = oh;
  if (p_lo < P) stage_out[p_lo * 64 + lane] = ol;
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 convolution, stride 1, zero padding 1, CIN -> 256 channels, channels-last (the pixel decoder's output
// convolutions, segmentation.py:186-204,253-283).  Pixel-stationary: a wave owns 32 pixels (the MFMA columns) and all
// 256 output channels (8 accumulator tiles); K = 9 taps x CIN runs in 16-wide steps, ordered (vertical tap, channel
// chunk, horizontal tap) so that the three horizontal neighbours of a chunk are read in consecutive steps.
//   * The pixel's own operand -- 16 input channels of the tap's neighbour -- goes from global memory straight into
//     registers LX steps ahead (32 contiguous bytes per lane; out-of-image taps read a block of zeros instead) and is
//     split into fp16 hi/lo between the previous step's MFMAs: the activations never pass through LDS.
//   * The weight fragments of a step (8 channel tiles x hi/lo = 16 pieces of 1 KiB, pre-split and pre-ordered by
//     tce_conv3x3_pack_f32) are requested LW steps ahead into registers, each wave a quarter of the stage, written into
//     a 4-stage LDS ring two steps before use and read back by all waves as A fragments.
//   * One barrier per step, placed after tile 5: by then every fragment of the stage is in registers (the stage can be
//     refilled), and tiles 6 and 7 are issued behind the barrier so the LDS latency of the next step's first
//     fragments hides under them.
//   * All loads of the loop are issued by hand in a fixed order (per mid-step: the lane's two operand loads, then the
//     wave's four weight loads), which makes the counted vmcnt at a mid-step exact.  They are all REGISTER loads on
//     purpose: a first version streamed the weights with LDS-DMA (global_load_lds) next to register loads of the
//     operand, and on real (not L2-resident) activations the counted wait let operand registers be read before their
//     data had arrived -- register loads and LDS-DMA loads do not retire in issue order relative to each other, so one
//     vmcnt cannot cover both.  (The FFN / row-linear kernels above count only DMA against DMA.)
// Ablations (tools/conv3_ablate.py, profiles/r02_conv3x3.txt): MFMA + LDS alone run at 1.4 PFLOP/s issued; the loads
// cost as much again because the L2 -> CU traffic (every workgroup streams the 2.4 MB of weights, every tap re-reads
// its pixels) is throughput-bound at the clock the MFMAs leave.
// ---------------------------------------------------------------------------------------------------------------
constexpr int CONV_PAD_STAGES = 6;  // = the kernel's weight lead LW

struct ConvArgs {
  const float* x;
  const unsigned char* wpk;
  const float* bias;
  float* out;
  long long ldx, ldo;
  int H, W, M;
  int* range_flag;
  int single;
  int m_off;  // first pixel of this launch (a launch may cover a row range [m_off, ...) of the map: the mixed form below)
};

// Ablation builds (tools/conv3_ablate.py, -DCONV_ABL=n; results are then wrong): bit 0: one workgroup per CU; bit 1: no
// barrier inside the loop; bit 2: no MFMA (fragments kept live); bit 3: no weight DMA inside the loop; bit 4: no operand
// loads inside the loop.
#ifndef CONV_ABL
#define CONV_ABL 0
#endif

template <int BYTE_OFF>
__device__ __forceinline__ void gload16(f32x4& dst, const float* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(BYTE_OFF) : "memory");
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// WAVES = 8 (round 5, A/B): 256 pixels per workgroup, two waves per SIMD sharing ONE weight ring -- half the weight bytes per pixel
// through L2 -> CU, which is what bounds the kernel; each wave then moves 2 of a stage's 16 pieces and keeps shorter register rings
// (XS 4, NW 3) to stay inside 256 registers.
template <int CIN, bool SINGLE, int WAVES = 4>
__global__ void __launch_bounds__(64 * WAVES, WAVES / 4) conv3x3_kernel(const ConvArgs p) {
  constexpr int NTL = 8;
  constexpr int PW = 16 / WAVES;          // weight pieces a wave moves per stage
  constexpr int STAGE = 2 * NTL * PIECE;  // one k-step of weights: 16 KiB
  constexpr int R = 4;                    // LDS ring stages
  constexpr int XS = WAVES == 4 ? 6 : 4, LX = XS + 1;  // operand ring slots (registers); the operand of step q is requested at mid-step q-LX
  constexpr int NW = WAVES == 4 ? CONV_PAD_STAGES - 2 : 3, LW = NW + 2;  // weight stages in flight (registers); those of step s are requested at mid-step s-LW
  constexpr int INFLIGHT = (2 + PW) * (LW - 3);  // loads younger than the weights a mid-step waits for (2 operand + PW weight per mid-step)
  static_assert(LW <= CONV_PAD_STAGES, "the packed stream's zero padding covers the weight lead");
  constexpr int BODY = 12;                // unrolled steps: 4 channel chunks x 3 horizontal taps (lcm of 3 and R)
  constexpr int ITERS = 3 * (CIN / 64);   // (vertical tap, group of 4 chunks)
  constexpr int KS = BODY * ITERS;
  static_assert(CIN % 64 == 0, "channel chunks come in groups of four");
  static_assert(BODY % R == 0 && BODY % XS == 0 && BODY % NW == 0 && LX >= LW, "ring positions are compile-time");
  static_assert(WAVES * WT_BYTES <= R * STAGE, "the epilogue's staging tiles alias the ring");
  __shared__ __attribute__((aligned(16))) unsigned char smem[R * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hf = lane >> 5;
  const int m0 = p.m_off + blockIdx.x * (32 * WAVES) + wave * 32;

  // this lane's pixel; per (vertical tap, chunk group) the addresses of its three horizontal neighbours' channels, or
  // of the zero block behind the weights for taps outside the image
  const int m = m0 + (lane & 31);
  const int px = m % p.W, py = (m / p.W) % p.H;
  const float* const zeros = reinterpret_cast<const float*>(p.wpk + (long long)(KS + CONV_PAD_STAGES) * STAGE) + 8 * hf;
  const float* const xme = p.x + (long long)min(m, p.M - 1) * p.ldx + 8 * hf;
  auto iter_ptrs = [&](const int it, const float* (&q)[3]) {
    const int dy = it / (CIN / 64) - 1, cq = it % (CIN / 64);
    const bool row_ok = it < ITERS && m < p.M && (unsigned)(py + dy) < (unsigned)p.H;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const bool ok = row_ok && (unsigned)(px + d - 1) < (unsigned)p.W;
      q[d] = (ok ? xme + (long long)(dy * p.W + d - 1) * p.ldx : zeros) + 64 * cq;
    }
  };

  // weights: this wave moves pieces PW*wave .. PW*wave+PW-1 of every stage (one float4 per lane and piece) through
  // registers into the ring
  const float* wp = reinterpret_cast<const float*>(p.wpk + (long long)(PW * wave) * PIECE + lane * 16);
  unsigned char* const wdst = smem + (PW * wave) * PIECE + lane * 16;
  f32x4 wr[NW][PW];
  auto wload = [&](f32x4 (&dst)[PW]) {
    if (!(CONV_ABL & 8)) {
      gload16<0>(dst[0], wp);
      gload16<PIECE>(dst[1], wp);
      if constexpr (PW == 4) {
        gload16<2 * PIECE>(dst[2], wp);
        gload16<3 * PIECE>(dst[3], wp);
      }
    }
    wp += STAGE / 4;
  };
  auto wstore = [&](const f32x4 (&src)[PW], const int stage) {
#pragma unroll
    for (int q = 0; q < PW; ++q) *reinterpret_cast<f32x4*>(wdst + stage * STAGE + q * PIECE) = src[q];
  };

  f32x4 xr[XS][2];
  const float* cur[3];
  const float* nxt[3];
  iter_ptrs(0, cur);
  iter_ptrs(1, nxt);
  // operand of body step JJ (>= BODY: of the next iteration) -> ring slot
  auto xload = [&](auto jjc, f32x4 (&dst)[2]) {
    constexpr int JJ = decltype(jjc)::value;
    constexpr int J = JJ % BODY;
    const float* const src = JJ < BODY ? cur[J % 3] : nxt[J % 3];
    if (CONV_ABL & 16) return;
    gload16<64 * (J / 3)>(dst[0], src);
    gload16<64 * (J / 3) + 16>(dst[1], src);
  };
  // "every load up to n operations ago has landed"; the register tuples named here are what the following code reads
  auto wait_loads = [&](auto nc, f32x4 (&xs)[2], f32x4 (&ws)[PW]) {
    constexpr int N = (CONV_ABL & 24) ? 0 : decltype(nc)::value;
    if constexpr (PW == 4) {
      asm volatile("s_waitcnt vmcnt(%6)"
                   : "+v"(xs[0]), "+v"(xs[1]), "+v"(ws[0]), "+v"(ws[1]), "+v"(ws[2]), "+v"(ws[3])
                   : "n"(N)
                   : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xs[0]), "+v"(xs[1]), "+v"(ws[0]), "+v"(ws[1]) : "n"(N) : "memory");
    }
  };
  using nfl = std::integral_constant<int, INFLIGHT>;

  // prologue = mid-steps -LX .. -1 in the loop's own order (wait + ring write of step m+2, then the requests)
  f32x16 acc[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  h16x8 bh, bl;
  static_for<0, LX>([&](auto mc) {
    constexpr int m = decltype(mc)::value - LX;
    if constexpr (m + 2 >= 0) {
      wait_loads(nfl{}, xr[(m + 2) % XS], wr[(m + 2) % NW]);
      wstore(wr[(m + 2) % NW], (m + 2) % R);
      if constexpr (m + 2 == 0) {  // operand 0 (two mid-steps older than these weights) feeds step 0 directly
        const float f[8] = {xr[0][0][0], xr[0][0][1], xr[0][0][2], xr[0][0][3], xr[0][1][0], xr[0][1][1], xr[0][1][2], xr[0][1][3]};
        const HL b = split8(f, SINGLE ? 1 : 0);
        bh = b.hi;
        bl = b.lo;
        asm volatile("" : "+v"(bh), "+v"(bl));  // the split reads xr[0] before a later request reuses it
      }
    }
    if constexpr (m + LX >= 0) xload(std::integral_constant<int, m + LX>{}, xr[(m + LX) % XS]);
    if constexpr (m + LW >= 0) wload(wr[(m + LW) % NW]);
  });
  __syncthreads();
  h16x8 fh[4], fl[4];
  {
    const unsigned char* const st = smem + lane * 16;
    fh[0] = *reinterpret_cast<const h16x8*>(st);
    fl[0] = *reinterpret_cast<const h16x8*>(st + PIECE);
    fh[1] = *reinterpret_cast<const h16x8*>(st + 2 * PIECE);
    fl[1] = *reinterpret_cast<const h16x8*>(st + 3 * PIECE);
  }

  for (int it = 0; it < ITERS; ++it) {
    static_for<0, BODY>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int nslot = (j + 1) % XS;  // operand of the next step: split under this step's first four tiles
      const unsigned char* const st = smem + (j % R) * STAGE + lane * 16;
      unsigned nh[4], nl[4];
      auto tile = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
#if CONV_ABL & 4
        const h16x8 k0 = fh[t % 4], k1 = fl[t % 4], k2 = bh, k3 = bl;
        asm volatile("" : : "v"(k0), "v"(k1), "v"(k2), "v"(k3));
#else
        if constexpr (!SINGLE) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[t % 4], bl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[t % 4], bh, acc[t], 0, 0, 0);
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[t % 4], bh, acc[t], 0, 0, 0);
#endif
      };
      // tiles 0..5, the fragments two tiles ahead of their MFMAs (tiles 0 and 1 were fetched by the previous step)
      static_for<0, 6>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        fh[(t + 2) % 4] = *reinterpret_cast<const h16x8*>(st + (2 * t + 4) * PIECE);
        fl[(t + 2) % 4] = *reinterpret_cast<const h16x8*>(st + (2 * t + 5) * PIECE);
        __builtin_amdgcn_sched_barrier(0);
        tile(tc);
        if constexpr (t < 4) {
          const float a = xr[nslot][t >> 1][2 * (t & 1)], b = xr[nslot][t >> 1][2 * (t & 1) + 1];
          if constexpr (SINGLE) {
            nh[t] = __builtin_bit_cast(unsigned, fp16x2_t{(__fp16)a, (__fp16)b});
            nl[t] = 0u;
          } else {
            const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(a, b);
            nh[t] = __builtin_bit_cast(unsigned, h);
            nl[t] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a - (float)h[0], b - (float)h[1]));
          }
          asm volatile("" : "+v"(nh[t]), "+v"(nl[t]));  // keeps the split here (it is only consumed by the next step)
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      // Mid-step: every fragment of this stage is in registers.  The weights of step j+2 (requested LW-2 mid-steps ago)
      // go into their stage -- free since the barrier of mid-step j-2 -- and the barrier publishes the stage written
      // one mid-step ago and releases this one.  Tiles 6 and 7 are issued AFTER the barrier so that the LDS latency
      // of the next step's first fragments hides under them.
      wait_loads(nfl{}, xr[(j + 2) % XS], wr[(j + 2) % NW]);
      wstore(wr[(j + 2) % NW], (j + 2) % R);
      if (!(CONV_ABL & 2)) __syncthreads();
      xload(std::integral_constant<int, j + LX>{}, xr[(j + LX) % XS]);
      wload(wr[(j + LW) % NW]);
      {
        const unsigned char* const sn = smem + ((j + 1) % R) * STAGE + lane * 16;
        fh[0] = *reinterpret_cast<const h16x8*>(sn);
        fl[0] = *reinterpret_cast<const h16x8*>(sn + PIECE);
        fh[1] = *reinterpret_cast<const h16x8*>(sn + 2 * PIECE);
        fl[1] = *reinterpret_cast<const h16x8*>(sn + 3 * PIECE);
      }
      __builtin_amdgcn_sched_barrier(0);
      tile(std::integral_constant<int, 6>{});
      tile(std::integral_constant<int, 7>{});
      __builtin_amdgcn_sched_barrier(0);
      bh = __builtin_bit_cast(h16x8, u32x4{nh[0], nh[1], nh[2], nh[3]});
      bl = __builtin_bit_cast(h16x8, u32x4{nl[0], nl[1], nl[2], nl[3]});
    });
#pragma unroll
    for (int d = 0; d < 3; ++d) cur[d] = nxt[d];
    iter_ptrs(it + 2, nxt);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (padding) loads
  __syncthreads();

  float* const wt = reinterpret_cast<float*>(smem + wave * WT_BYTES);
  tce_amax_t amax = 0;
  const ResTile none = {};
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    tile_bias_res<0>(acc[t], tile_bias_load(p.bias, 32 * t, lane), none, wt, lane);
    tile_store(acc[t], p.out, p.ldo, m0, p.M, 32 * t, wt, lane, amax);
  }
  tce_range_report(p.range_flag, amax);
}

// w [256, 9*CIN] (k = tap*CIN + c) -> 9*CIN/16 + 4 stages of 16 pieces in the kernel's K order: stage s = 12*it + j is
// vertical tap dy = it / (CIN/64), channel chunk cc = 4*(it % (CIN/64)) + j/3, horizontal tap dx = j % 3 (the three
// horizontal neighbours of a chunk in consecutive steps: they share cache lines).  Piece 2t (+1) = hi (lo) fragment of
// channel tile t: lane (r, hf) holds w[32t + r][(3dy+dx)*CIN + 16cc + 8hf + 0..7].  CONV_PAD_STAGES trailing stages of
// zeros (the weight requests run that many steps ahead) and 2 KiB of zeros that out-of-image taps read as their operand.
__global__ void __launch_bounds__(256) conv3x3_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ out,
                                                           const int CIN, const long long units, const int single) {
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  if (u >= units) return;
  const int lane = (int)(u & 63);
  const long long pg = u >> 6;
  const int piece = (int)(pg & 15);
  const long long s = pg >> 4;
  const int r = lane & 31, hf = lane >> 5, t = piece >> 1;
  u32x4 o = {0u, 0u, 0u, 0u};
  if (s < 9 * CIN / 16) {
    const int it = (int)(s / 12), j = (int)(s % 12);
    const int dy = it / (CIN / 64), cc = 4 * (it % (CIN / 64)) + j / 3, dx = j % 3;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w[(long long)(32 * t + r) * (9 * CIN) + (3 * dy + dx) * CIN + 16 * cc + 8 * hf + e];
    const HL f = split8(v, single);
    o = __builtin_bit_cast(u32x4, (piece & 1) ? f.lo : f.hi);
  }
  reinterpret_cast<u32x4*>(out)[u] = o;
}

inline bool conv3x3_shape_ok(int Cin, int N) { return Cin == 256 && N == 256; }
inline long long conv3x3_units(int Cin) { return (long long)(9 * Cin / 16 + CONV_PAD_STAGES) * 16 * 64 + 128; }

// ---------------------------------------------------------------------------------------------------------------
// Swin patch embedding (swin_transformer.py:433-471: Conv2d(3, C, 4, stride 4) + LayerNorm) on the matrix cores: a wave
// takes 32 patches as MFMA columns, K = 3 x 4 x 4 = 48 taps in three 16-wide steps whose order puts one input channel
// per step (k = 16c + 4ky + kx), so a lane reads two 16-byte row segments of its patch per step straight into the
// operand registers; the C x 48 weight is split into fragments once per wave and stays in registers.  Bias, LayerNorm
// over the C channels a lane pair holds, and full-line stores.  HBM-bound (reads the clip once, writes [tokens, C]).
// ---------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(256) patch_embed_mfma_kernel(const float* __restrict__ frames, const float* __restrict__ w,
                                                               const float* __restrict__ b, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ out,
                                                               const int H, const int W, const float eps, const int ntok,
                                                               const int Hp, const int Wp, int* const range_flag,
                                                               const int single) {
  constexpr int C = 32 * NT;
  __shared__ __attribute__((aligned(16))) unsigned char wtbuf[4 * WT_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hf = lane >> 5;
  const int m0 = blockIdx.x * 128 + wave * 32;
  if (m0 >= ntok) return;  // no workgroup-wide barrier below
  float* const wt = reinterpret_cast<float*>(wtbuf + wave * WT_BYTES);

  // this lane's patch: rows 4py + 2hf + {0,1} of every channel
  const int tok = min(m0 + r, ntok - 1);
  const int t = tok / (Hp * Wp), rem = tok - t * (Hp * Wp);
  const int py = rem / Wp, px = rem - py * Wp;
  const bool vec = ((W & 3) == 0) && (px * 4 + 3 < W);
  f32x4 raw[3][2];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int yy = py * 4 + 2 * hf + q;
      const float* const row = frames + (((long long)t * 3 + c) * H + min(yy, H - 1)) * W;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (yy < H) {
        if (vec) {
          v = *reinterpret_cast<const f32x4*>(row + px * 4);
        } else {
#pragma unroll
          for (int kx = 0; kx < 4; ++kx)
            if (px * 4 + kx < W) v[kx] = row[px * 4 + kx];
        }
      }
      raw[c][q] = v;
    }
  // weight fragments: lane (r, hf) holds w[32t + r][16s + 8hf + 0..7]
  h16x8 ah[NT][3], al[NT][3];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt)
#pragma unroll
    for (int sK = 0; sK < 3; ++sK) {
      const float* const pw = w + (long long)(32 * tt + r) * 48 + 16 * sK + 8 * hf;
      const f32x4 a = *reinterpret_cast<const f32x4*>(pw), c4 = *reinterpret_cast<const f32x4*>(pw + 4);
      const float f[8] = {a[0], a[1], a[2], a[3], c4[0], c4[1], c4[2], c4[3]};
      const HL sp = split8(f, single);
      ah[tt][sK] = sp.hi;
      al[tt][sK] = sp.lo;
    }
  f32x16 acc[NT];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[tt][i] = 0.f;
#pragma unroll
  for (int sK = 0; sK < 3; ++sK) {
    const float f[8] = {raw[sK][0][0], raw[sK][0][1], raw[sK][0][2], raw[sK][0][3],
                        raw[sK][1][0], raw[sK][1][1], raw[sK][1][2], raw[sK][1][3]};
    const HL x = split8(f, single);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      if (!single) {
        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tt][sK], x.lo, acc[tt], 0, 0, 0);
        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tt][sK], x.hi, acc[tt], 0, 0, 0);
      }
      acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tt][sK], x.hi, acc[tt], 0, 0, 0);
    }
  }
  const ResTile none = {};
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) tile_bias_res<0>(acc[tt], tile_bias_load(b, 32 * tt, lane), none, wt, lane);
  rows_layernorm<NT>(acc, gamma, beta, eps, hf);
  tce_amax_t amax = 0;
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) tile_store(acc[tt], out, C, m0, ntok, 32 * tt, wt, lane, amax);
  tce_range_report(range_flag, amax);
}

}  // namespace

// The library builds this file as TWO translation units (chain_ffn.hip: TCE_CHAIN_PART 1, chain_rowlin.hip: TCE_CHAIN_PART 2) so that
// the fused-FFN family and the token-stationary linear family -- each a few dozen instantiations -- compile in parallel; the
// kernel templates above are shared text, only the entry points (= the instantiation sites) are divided.  Compiled directly
// (tools/ffn_ablate.py) it is one unit with everything.
#ifndef TCE_CHAIN_PART
#define TCE_CHAIN_PART 0
#endif
int tce_chain_rowlin_set_stamps(long long* dev_buf);

#if TCE_CHAIN_PART != 2
extern "C" int tce_debug_ffn_set_stamp_buffer(long long* dev_buf) {
#if TCE_CHAIN_PART == 1
  if (tce_chain_rowlin_set_stamps(dev_buf) != 0) return -1;  // the other translation unit's copy of the pointer
#endif
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
                     (unsigned char*)packed, C, Hd, ffn_pieces(C), units, tce_gemm_single_pass());
  TCE_CHECK_LAUNCH("tce_ffn_pack_f32");
  return TCE_OK;
}

extern "C" int tce_ffn_pack_batched_f32(const float* W1, const float* b1, const float* W2, void* packed, int32_t C, int32_t Hd,
                                        int32_t batch, tceStream stream) {
  TCE_CHECK_ARG(ffn_shape_ok(C, Hd) && batch > 0, "tce_ffn_pack_batched_f32: unsupported shape C=%d hidden=%d batch=%d", C, Hd, batch);
  TCE_CHECK_ARG(W1 && W2 && packed && tce_aligned16(packed), "tce_ffn_pack_batched_f32: null / misaligned pointer");
  const long long units = ffn_units(C, Hd);
  hipLaunchKernelGGL(ffn_pack_kernel, dim3(tce_cdiv(units, 256), batch), dim3(256), 0, (hipStream_t)stream, W1, b1, W2,
                     (unsigned char*)packed, C, Hd, ffn_pieces(C), units, tce_gemm_single_pass());
  TCE_CHECK_LAUNCH("tce_ffn_pack_batched_f32");
  return TCE_OK;
}

extern "C" int tce_ffn_pack_chain_f32(const float* W1, const float* b1, const float* W2, void* packed, int32_t C, int32_t Hd,
                                      tceStream stream) {
  TCE_CHECK_ARG(C == 256 && ffn_shape_ok(C, Hd), "tce_ffn_pack_chain_f32: C must be 256, hidden %% 32 == 0 (C=%d hidden=%d)", C, Hd);
  TCE_CHECK_ARG(W1 && W2 && packed && tce_aligned16(packed), "tce_ffn_pack_chain_f32: null / misaligned pointer");
  const long long units = ffn_units(C, Hd);
  hipLaunchKernelGGL(ffn_pack_kernel, dim3(tce_cdiv(units, 256)), dim3(256), 0, (hipStream_t)stream, W1, b1, W2,
                     (unsigned char*)packed, C, Hd, ffn_pieces(C), units, tce_gemm_single_pass(), 1);
  TCE_CHECK_LAUNCH("tce_ffn_pack_chain_f32");
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
  FfnArgs a = {};
  a.x = x; a.wpk = (const unsigned char*)packed; a.b2 = b2; a.res_mode = 1;
  a.g_in = g_in; a.be_in = be_in; a.g_out = g_out; a.be_out = be_out;
  a.out = out; a.ldx = ldx; a.ldo = ldo; a.M = M; a.NI = Hd / 32 + 1; a.eps_in = eps_in; a.eps_out = eps_out; a.range_flag = tce_range_flag(); a.single = tce_gemm_single_pass();
  a.wdiv = 1;
  hipStream_t s = (hipStream_t)stream;
  if (C == 256) ffn_launch<256, 4>(a, act, s);
  else if (C == 192) ffn_launch<192, 4>(a, act, s);
  else if (C == 128) ffn_launch<128, 8>(a, act, s);
  else ffn_launch<96, 8>(a, act, s);
  TCE_CHECK_LAUNCH("tce_ffn_fused_f32");
  return TCE_OK;
}

extern "C" int tce_debug_ffn_set_half(int32_t mode) {
  TCE_CHECK_ARG(mode >= -1 && mode <= 1, "tce_debug_ffn_set_half: -1 (never), 0 (automatic) or 1 (always)");
  g_ffn_half = mode;
  return TCE_OK;
}

extern "C" int64_t tce_ffn_split_ws_floats(int32_t M, int32_t C, int32_t Hd, int32_t act) {
  if (M <= 0 || !ffn_shape_ok(C, Hd) || act != 1) return 0;
  return ffn_split_plan(M, C, Hd, 1).ws_floats;
}

extern "C" int32_t tce_ffn_split_counters(int32_t M, int32_t C, int32_t Hd, int32_t act) {
  if (M <= 0 || !ffn_shape_ok(C, Hd) || act != 1) return 0;
  return ffn_split_plan(M, C, Hd, 1).nblk;
}

extern "C" int tce_ffn_fused_split_f32(const float* x, int64_t ldx, const void* packed, const float* b2, const float* g_in,
                                       const float* be_in, float eps_in, const float* g_out, const float* be_out, float eps_out,
                                       float* out, int64_t ldo, int32_t M, int32_t C, int32_t Hd, int32_t act, float* ws,
                                       int64_t ws_floats, int32_t* counters, int32_t n_counters, tceStream stream) {
  TCE_CHECK_ARG(ffn_shape_ok(C, Hd), "tce_ffn_fused_split_f32: unsupported shape C=%d hidden=%d", C, Hd);
  TCE_CHECK_ARG(M > 0 && x && packed && b2 && out, "tce_ffn_fused_split_f32: null pointer or M <= 0");
  TCE_CHECK_ARG(act == 1, "tce_ffn_fused_split_f32: act must be 1 (ReLU)");
  TCE_CHECK_ARG(ldx >= C && ldo >= C && ldx % 4 == 0 && ldo % 4 == 0, "tce_ffn_fused_split_f32: bad row pitch");
  TCE_CHECK_ARG(tce_aligned16(x) && tce_aligned16(out) && tce_aligned16(packed) && tce_aligned16(b2),
                "tce_ffn_fused_split_f32: x/out/packed/b2 must be 16-byte aligned");
  TCE_CHECK_ARG((!g_in || (be_in && tce_aligned16(g_in) && tce_aligned16(be_in))) &&
                    (!g_out || (be_out && tce_aligned16(g_out) && tce_aligned16(be_out))),
                "tce_ffn_fused_split_f32: LayerNorm gamma/beta must come in pairs, 16-byte aligned");
  const FfnSplitPlan plan = ffn_split_plan(M, C, Hd, 1);
  TCE_CHECK_ARG(plan.p > 0, "tce_ffn_fused_split_f32: no split is planned for M=%d C=%d hidden=%d (tce_ffn_split_ws_floats is 0): "
                "call tce_ffn_fused_f32", M, C, Hd);
  TCE_CHECK_ARG(ws && tce_aligned16(ws) && ws_floats >= plan.ws_floats && counters && n_counters >= plan.nblk,
                "tce_ffn_fused_split_f32: workspace of %lld floats and %d zeroed counters needed", plan.ws_floats, plan.nblk);
  TCE_CHECK_ARG(!(ws <= out + (long long)(M - 1) * ldo + C - 1 && out <= ws + plan.ws_floats - 1) &&
                    !(ws <= x + (long long)(M - 1) * ldx + C - 1 && x <= ws + plan.ws_floats - 1),
                "tce_ffn_fused_split_f32: the workspace overlaps x / out");
  FfnArgs a = {};
  a.x = x; a.wpk = (const unsigned char*)packed; a.b2 = b2; a.res_mode = 1;
  a.g_in = g_in; a.be_in = be_in; a.g_out = g_out; a.be_out = be_out;
  a.out = out; a.ldx = ldx; a.ldo = ldo; a.M = M; a.NI = Hd / 32 + 1; a.eps_in = eps_in; a.eps_out = eps_out; a.range_flag = tce_range_flag(); a.single = tce_gemm_single_pass();
  a.wdiv = 1;
  a.ws = ws; a.cnt = counters; a.sp_p = plan.p; a.sp_k0 = plan.k[0]; a.sp_k1 = plan.k[1]; a.nblk = plan.nblk;
  ffn_launch<256, 4>(a, act, (hipStream_t)stream);
  TCE_CHECK_LAUNCH("tce_ffn_fused_split_f32");
  return TCE_OK;
}

extern "C" int tce_xattn_prepare_f32(const float* k, const float* v, const float* wqT_ext, const float* wo, float* W1, float* b1,
                                     float* W2, int32_t L, int32_t group, int32_t batch, tceStream stream) {
  TCE_CHECK_ARG(k && v && wqT_ext && wo && W1 && b1 && W2, "tce_xattn_prepare_f32: null pointer");
  TCE_CHECK_ARG((group == 32 || group == 8) && L >= 1 && L <= group && batch >= 1,
                "tce_xattn_prepare_f32: group must be 32 or 8 and 1 <= L <= group (L=%d group=%d)", L, group);
  TCE_CHECK_ARG(tce_aligned16(k) && tce_aligned16(v) && tce_aligned16(wqT_ext) && tce_aligned16(wo),
                "tce_xattn_prepare_f32: pointers must be 16-byte aligned");
  const int Hd = 8 * group;
  hipLaunchKernelGGL(xattn_prepare_kernel, dim3(tce_cdiv(2 * Hd * 256 + Hd, 256), batch), dim3(256), 0, (hipStream_t)stream,
                     k, v, wqT_ext, wo, W1, b1, W2, L, group);
  TCE_CHECK_LAUNCH("tce_xattn_prepare_f32");
  return TCE_OK;
}

extern "C" int tce_xattn_pack_f32(const float* k, const float* v, const float* wqT_ext, const float* wo, void* packed, int32_t L,
                                  int32_t group, int32_t batch, tceStream stream) {
  TCE_CHECK_ARG(k && v && wqT_ext && wo && packed, "tce_xattn_pack_f32: null pointer");
  TCE_CHECK_ARG((group == 32 || group == 8) && L > 0 && L <= group && batch > 0, "tce_xattn_pack_f32: group must be 32 or 8, 0 < L <= group");
  TCE_CHECK_ARG(tce_aligned16(k) && tce_aligned16(v) && tce_aligned16(wqT_ext) && tce_aligned16(wo) && tce_aligned16(packed),
                "tce_xattn_pack_f32: pointers must be 16-byte aligned");
  const int Hd = 8 * group;
  const long long units = ffn_units(256, Hd);
  const int P = ffn_pieces(256), SL = 1 + P / 2;           // slot 0 = piece 0, slot sl = pieces 2 sl - 1, 2 sl
  const long long threads = (long long)(Hd / 32 + 2) * SL * 64;
  hipLaunchKernelGGL(xattn_pack_fused_kernel, dim3(tce_cdiv(threads, 256), batch), dim3(256), 0, (hipStream_t)stream, k, v, wqT_ext,
                     wo, (unsigned char*)packed, L, group, P, SL, threads, units, tce_gemm_single_pass());
  TCE_CHECK_LAUNCH("tce_xattn_pack_f32");
  return TCE_OK;
}

static int xattn_launch(const tceXattnArgs* args, const tceXattnFfnArgs* ffn, tceStream stream);

extern "C" int tce_xattn_fused_f32(const tceXattnArgs* args, tceStream stream) { return xattn_launch(args, nullptr, stream); }

// cross-attention -> FFN as ONE launch (round 5): see tceXattnFfnArgs in the header
extern "C" int tce_xattn_ffn_fused_f32(const tceXattnArgs* args, const tceXattnFfnArgs* ffn, tceStream stream) {
  TCE_CHECK_ARG(args && ffn, "tce_xattn_ffn_fused_f32: null args");
  TCE_CHECK_ARG(ffn->packed && ffn->b2 && ffn->mid && ffn->hidden > 0 && ffn->hidden % 32 == 0 && ffn_shape_ok(256, ffn->hidden),
                "tce_xattn_ffn_fused_f32: bad FFN stage (packed / b2 / mid / hidden)");
  TCE_CHECK_ARG(ffn->act == 1, "tce_xattn_ffn_fused_f32: the FFN stage's activation must be 1 (ReLU)");
  TCE_CHECK_ARG(ffn->ldmid >= 256 && ffn->ldmid % 4 == 0 && ffn->sMid % 4 == 0 && tce_aligned16(ffn->mid) && tce_aligned16(ffn->packed) &&
                    tce_aligned16(ffn->b2) && (!ffn->g_out || (ffn->be_out && tce_aligned16(ffn->g_out) && tce_aligned16(ffn->be_out))),
                "tce_xattn_ffn_fused_f32: FFN stage pitch / alignment");
  TCE_CHECK_ARG(args->res_mode == 1, "tce_xattn_ffn_fused_f32: the attention stage's residual must be additive");
  TCE_CHECK_ARG(ffn->mid != args->x && ffn->mid != args->out && ffn->mid != args->res,
                "tce_xattn_ffn_fused_f32: mid must be a buffer of its own (it is written and re-read inside the launch)");
  return xattn_launch(args, ffn, stream);
}

static int xattn_launch(const tceXattnArgs* args, const tceXattnFfnArgs* ffn, tceStream stream) {
  TCE_CHECK_ARG(args != nullptr, "tce_xattn_fused_f32: null args");
  const tceXattnArgs& q = *args;
  TCE_CHECK_ARG(q.M > 0 && q.x && q.packed && q.bo && q.out, "tce_xattn_fused_f32: null pointer or M <= 0");
  TCE_CHECK_ARG(q.res_mode == 1 || q.res_mode == 2, "tce_xattn_fused_f32: res_mode must be 1 (add) or 2 (multiply)");
  TCE_CHECK_ARG(q.group == 32 || q.group == 8, "tce_xattn_fused_f32: group (key slots per head) must be 32 or 8");
  TCE_CHECK_ARG(q.ldx >= 256 && q.ldo >= 256 && q.ldx % 4 == 0 && q.ldo % 4 == 0 && (!q.res || (q.ldres >= 256 && q.ldres % 4 == 0)) &&
                    (!q.a2 || (q.lda2 >= 256 && q.lda2 % 4 == 0)) && q.sX % 4 == 0 && q.sRes % 4 == 0 && q.sOut % 4 == 0,
                "tce_xattn_fused_f32: bad row pitch / batch stride");
  TCE_CHECK_ARG(tce_aligned16(q.x) && tce_aligned16(q.out) && tce_aligned16(q.packed) && tce_aligned16(q.bo) &&
                    (!q.a2 || tce_aligned16(q.a2)) && (!q.res || tce_aligned16(q.res)),
                "tce_xattn_fused_f32: pointers must be 16-byte aligned");
  TCE_CHECK_ARG(!q.g_out || (q.be_out && tce_aligned16(q.g_out) && tce_aligned16(q.be_out)),
                "tce_xattn_fused_f32: LayerNorm gamma/beta must come in pairs, 16-byte aligned");
  FfnArgs a = {};
  a.x = q.x; a.wpk = (const unsigned char*)q.packed; a.b2 = q.bo; a.g_out = q.g_out; a.be_out = q.be_out; a.out = q.out;
  a.ldx = q.ldx; a.ldo = q.ldo; a.a2 = q.a2; a.lda2 = q.lda2; a.a2_rows = q.a2_rows; a.res = q.res; a.ldres = q.ldres;
  a.res_mode = q.res_mode; a.sX = q.sX; a.sRes = q.sRes; a.sOut = q.sOut; a.sW = q.sW;
  a.M = q.M; a.NI = (8 * q.group) / 32 + 1; a.eps_out = q.eps_out; a.range_flag = tce_range_flag(); a.single = tce_gemm_single_pass();
  a.wdiv = q.w_div > 0 ? q.w_div : 1;
  const int batch = q.batch > 0 ? q.batch : 1;
  const dim3 grid(tce_cdiv(a.M, 128), batch), block(256);
  if (ffn) {
    a.wpk2 = (const unsigned char*)ffn->packed; a.b22 = ffn->b2; a.g_out2 = ffn->g_out; a.be_out2 = ffn->be_out; a.eps_out2 = ffn->eps_out;
    a.mid = ffn->mid; a.ldmid = ffn->ldmid; a.sMid = ffn->sMid; a.NI2 = ffn->hidden / 32 + 1;
    if (a.single) {
      if (q.group == 32) hipLaunchKernelGGL((ffn_fused_kernel<256, 4, 3, true, 1>), grid, block, 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((ffn_fused_kernel<256, 4, 4, true, 1>), grid, block, 0, (hipStream_t)stream, a);
    } else {
      if (q.group == 32) hipLaunchKernelGGL((ffn_fused_kernel<256, 4, 3, false, 1>), grid, block, 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((ffn_fused_kernel<256, 4, 4, false, 1>), grid, block, 0, (hipStream_t)stream, a);
    }
    TCE_CHECK_LAUNCH("tce_xattn_ffn_fused_f32");
    return TCE_OK;
  }
  if (a.single) {
    if (q.group == 32) hipLaunchKernelGGL((ffn_fused_kernel<256, 4, 3, true>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((ffn_fused_kernel<256, 4, 4, true>), grid, block, 0, (hipStream_t)stream, a);
  } else {
    if (q.group == 32) hipLaunchKernelGGL((ffn_fused_kernel<256, 4, 3, false>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((ffn_fused_kernel<256, 4, 4, false>), grid, block, 0, (hipStream_t)stream, a);
  }
  TCE_CHECK_LAUNCH("tce_xattn_fused_f32");
  return TCE_OK;
}

#endif  // TCE_CHAIN_PART != 2

#if TCE_CHAIN_PART != 1
int tce_chain_rowlin_set_stamps(long long* dev_buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_ffn_stamps), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : -1;
}

extern "C" int64_t tce_rowlin_packed_bytes(int32_t N, int32_t K) { return rowlin_shape_ok(N, K) ? rowlin_units(N, K) * 16 : -1; }

extern "C" int tce_rowlin_pack_f32(const float* W, int64_t ldw, void* packed, int32_t N, int32_t K, tceStream stream) {
  TCE_CHECK_ARG(rowlin_shape_ok(N, K), "tce_rowlin_pack_f32: unsupported shape N=%d K=%d (K in 96/128/192/256/384/512, N %% 32 == 0)", N, K);
  TCE_CHECK_ARG(W && packed && tce_aligned16(packed) && ldw >= K, "tce_rowlin_pack_f32: null / misaligned pointer or ldw < K");
  const long long units = rowlin_units(N, K);
  hipLaunchKernelGGL(rowlin_pack_kernel, dim3(tce_cdiv(units, 256)), dim3(256), 0, (hipStream_t)stream, W,
                     (unsigned char*)packed, N, K, (long long)ldw, units, tce_gemm_single_pass());
  TCE_CHECK_LAUNCH("tce_rowlin_pack_f32");
  return TCE_OK;
}

extern "C" int tce_rowlin_f32(const tceRowLinArgs* args, tceStream stream) {
  TCE_CHECK_ARG(args != nullptr, "tce_rowlin_f32: null args");
  const tceRowLinArgs& q = *args;
  TCE_CHECK_ARG(rowlin_shape_ok(q.N, q.K), "tce_rowlin_f32: unsupported shape N=%d K=%d", q.N, q.K);
  TCE_CHECK_ARG(q.M > 0 && q.x && q.packed && q.out, "tce_rowlin_f32: null pointer or M <= 0");
  TCE_CHECK_ARG(q.act >= 0 && q.act <= 2 && q.res_mode >= 0 && q.res_mode <= 2, "tce_rowlin_f32: bad act / res_mode");
  TCE_CHECK_ARG(q.res_mode == 0 || q.res, "tce_rowlin_f32: res_mode set without res");
  TCE_CHECK_ARG(q.ldx >= q.K && q.ldx % 4 == 0 && q.ldo >= q.N && q.ldo % 4 == 0 && (!q.res || (q.ldres >= q.N && q.ldres % 4 == 0)) &&
                    (!q.a2 || (q.lda2 >= q.K && q.lda2 % 4 == 0)),
                "tce_rowlin_f32: bad row pitch");
  TCE_CHECK_ARG(q.sX % 4 == 0 && q.sA2 % 4 == 0 && q.sRes % 4 == 0 && q.sOut % 4 == 0, "tce_rowlin_f32: batch strides must be multiples of 4 floats");
  TCE_CHECK_ARG(tce_aligned16(q.x) && tce_aligned16(q.out) && tce_aligned16(q.packed) && (!q.a2 || tce_aligned16(q.a2)) &&
                    (!q.res || tce_aligned16(q.res)) && (!q.bias || tce_aligned16(q.bias)),
                "tce_rowlin_f32: pointers must be 16-byte aligned");
  TCE_CHECK_ARG((!q.g_in || (q.be_in && tce_aligned16(q.g_in) && tce_aligned16(q.be_in))) &&
                    (!q.g_out || (q.be_out && tce_aligned16(q.g_out) && tce_aligned16(q.be_out))),
                "tce_rowlin_f32: LayerNorm gamma/beta must come in pairs, 16-byte aligned");
  TCE_CHECK_ARG(!q.g_out || (q.N == 256 && q.K <= 384), "tce_rowlin_f32: the output LayerNorm is built for N = 256, K <= 384");
  LinArgs a;
  a.x = q.x; a.a2 = q.a2; a.wpk = (const unsigned char*)q.packed; a.bias = q.bias; a.res = q.res; a.out = q.out;
  a.g_in = q.g_in; a.be_in = q.be_in; a.g_out = q.g_out; a.be_out = q.be_out;
  a.ldx = q.ldx; a.lda2 = q.lda2; a.ldres = q.ldres; a.ldo = q.ldo;
  a.sX = q.sX; a.sA2 = q.sA2; a.sRes = q.sRes; a.sOut = q.sOut;
  a.M = q.M; a.N = q.N; a.a2_rows = q.a2_rows; a.act = q.act; a.res_mode = q.res_mode;
  a.eps_in = q.eps_in; a.eps_out = q.eps_out; a.range_flag = tce_range_flag(); a.single = tce_gemm_single_pass();
  const int batch = q.batch > 0 ? q.batch : 1;
  const bool row = q.g_out != nullptr;
  hipStream_t s = (hipStream_t)stream;
  if (q.K == 512) rowlin_launch<512>(a, batch, row, s);  // Swin-B stage 3 (C = 512): x = 256 registers, one workgroup per CU
  else if (q.K == 384) rowlin_launch<384>(a, batch, row, s);  // Swin stage 3 (C = 384): x = 192 registers, one workgroup per CU
  else if (q.K == 256) rowlin_launch<256>(a, batch, row, s);
  else if (q.K == 192) rowlin_launch<192>(a, batch, row, s);
  else if (q.K == 128) rowlin_launch<128>(a, batch, row, s);
  else rowlin_launch<96>(a, batch, row, s);
  TCE_CHECK_LAUNCH("tce_rowlin_f32");
  return TCE_OK;
}

#endif  // TCE_CHAIN_PART != 1

#if TCE_CHAIN_PART != 2
extern "C" int64_t tce_conv3x3_packed_bytes(int32_t Cin, int32_t N) { return conv3x3_shape_ok(Cin, N) ? conv3x3_units(Cin) * 16 : -1; }

extern "C" int tce_conv3x3_pack_f32(const float* w, void* packed, int32_t Cin, int32_t N, tceStream stream) {
  TCE_CHECK_ARG(conv3x3_shape_ok(Cin, N), "tce_conv3x3_pack_f32: unsupported shape Cin=%d N=%d (256 -> 256)", Cin, N);
  TCE_CHECK_ARG(w && packed && tce_aligned16(packed), "tce_conv3x3_pack_f32: null / misaligned pointer");
  const long long units = conv3x3_units(Cin);
  hipLaunchKernelGGL(conv3x3_pack_kernel, dim3(tce_cdiv(units, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (unsigned char*)packed, Cin, units, tce_gemm_single_pass());
  TCE_CHECK_LAUNCH("tce_conv3x3_pack_f32");
  return TCE_OK;
}

static int g_conv3_waves = 0;
extern "C" int tce_debug_conv3x3_set_waves(int32_t waves) {
  TCE_CHECK_ARG(waves == 0 || waves == 4 || waves == 8, "tce_debug_conv3x3_set_waves: 0 (automatic), 4 or 8");
  g_conv3_waves = waves;
  return TCE_OK;
}

extern "C" int tce_conv3x3_f32(const float* x, int64_t ldx, const void* packed, const float* bias, float* out, int64_t ldo,
                               int32_t T, int32_t H, int32_t W, int32_t Cin, int32_t N, tceStream stream) {
  TCE_CHECK_ARG(conv3x3_shape_ok(Cin, N), "tce_conv3x3_f32: unsupported shape Cin=%d N=%d (256 -> 256)", Cin, N);
  TCE_CHECK_ARG(x && packed && out && T > 0 && H > 0 && W > 0 && (long long)T * H * W < (1ll << 31),
                "tce_conv3x3_f32: null pointer or bad sizes");
  TCE_CHECK_ARG(ldx >= Cin && ldx % 4 == 0 && ldo >= N && ldo % 4 == 0, "tce_conv3x3_f32: bad row pitch");
  TCE_CHECK_ARG(tce_aligned16(x) && tce_aligned16(out) && tce_aligned16(packed) && (!bias || tce_aligned16(bias)),
                "tce_conv3x3_f32: pointers must be 16-byte aligned");
  ConvArgs a;
  a.x = x; a.wpk = (const unsigned char*)packed; a.bias = bias; a.out = out;
  a.ldx = ldx; a.ldo = ldo; a.H = H; a.W = W; a.M = T * H * W;
  a.range_flag = tce_range_flag(); a.single = tce_gemm_single_pass();
  // 256-pixel (8-wave) workgroups halve the weight bytes per pixel; a round of them takes 1.75 x a round of 128-pixel workgroups
  // (measured, tools/conv3_bench.py: 128400 px 508 -> 444 us, 72000 px 333 -> 337, 18000 px 100 -> 142): taken when the rounds of
  // 256 workgroups that way cost less -- config 5 and clip groups, not the single config-2 clip.  TCE_CONV3_WAVES=4|8 / tce_debug_conv3x3_set_waves force one.
  static const int env_force = []() { const char* e = getenv("TCE_CONV3_WAVES"); return e ? atoi(e) : 0; }();
  const int force = g_conv3_waves ? g_conv3_waves : env_force;
  const int r4 = tce_cdiv(tce_cdiv(a.M, 128), 256), r8 = tce_cdiv(tce_cdiv(a.M, 256), 256);
  a.m_off = 0;
  // Mixed form (round 5): full rounds of 256-pixel workgroups, then the REMAINDER as 128-pixel workgroups (they run one wave per
  // SIMD: a round of them costs 1 against 1.75).  72000 px: 256 wide workgroups + 51 narrow ones = 1.75 + 1 rounds against 3 narrow
  // rounds (or 2 x 1.75 wide ones).  In units of a narrow round x 4:
  const int full8 = tce_cdiv(a.M, 256) / 256;                       // complete rounds of wide workgroups
  const int rem_px = a.M - full8 * 256 * 256;
  const int mixed4 = 7 * full8 + 4 * tce_cdiv(tce_cdiv(rem_px, 128), 256);
  if (!force && !a.single && full8 >= 1 && rem_px > 0 && mixed4 < 4 * r4 && mixed4 < 7 * r8) {
    hipLaunchKernelGGL((conv3x3_kernel<256, false, 8>), dim3(full8 * 256), dim3(512), 0, (hipStream_t)stream, a);
    a.m_off = full8 * 256 * 256;
    hipLaunchKernelGGL((conv3x3_kernel<256, false>), dim3(tce_cdiv(rem_px, 128)), dim3(256), 0, (hipStream_t)stream, a);
    TCE_CHECK_LAUNCH("tce_conv3x3_f32(mixed)");
    return TCE_OK;
  }
  const int wide = force ? force : (4 * r4 > 7 * r8 ? 8 : 4);
  if (wide == 8 && !a.single) hipLaunchKernelGGL((conv3x3_kernel<256, false, 8>), dim3(tce_cdiv(a.M, 256)), dim3(512), 0, (hipStream_t)stream, a);
  else if (a.single) hipLaunchKernelGGL((conv3x3_kernel<256, true>), dim3(tce_cdiv(a.M, 128)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((conv3x3_kernel<256, false>), dim3(tce_cdiv(a.M, 128)), dim3(256), 0, (hipStream_t)stream, a);
  TCE_CHECK_LAUNCH("tce_conv3x3_f32");
  return TCE_OK;
}

// split-fp16 / fp16 modes of tce_patch_embed_f32 (norm.hip dispatches here); false = shape not covered
bool tce_patch_embed_mfma(const float* frames, const float* w, const float* b, const float* gamma, const float* beta,
                          float* out, int H, int W, int C, float eps, long long ntok, int Hp, int Wp, hipStream_t s) {
  if ((C != 96 && C != 128 && C != 192) || ntok >= (1ll << 31) || !tce_aligned16(frames) || !tce_aligned16(out) ||
      !tce_aligned16(w) || !tce_aligned16(b) || !tce_aligned16(gamma) || !tce_aligned16(beta))
    return false;
  const dim3 grid(tce_cdiv(ntok, 128)), block(256);
  int* const rf = tce_range_flag();
  const int single = tce_gemm_single_pass();
  if (C == 96) hipLaunchKernelGGL((patch_embed_mfma_kernel<3>), grid, block, 0, s, frames, w, b, gamma, beta, out, H, W, eps, (int)ntok, Hp, Wp, rf, single);
  else if (C == 128) hipLaunchKernelGGL((patch_embed_mfma_kernel<4>), grid, block, 0, s, frames, w, b, gamma, beta, out, H, W, eps, (int)ntok, Hp, Wp, rf, single);
  else hipLaunchKernelGGL((patch_embed_mfma_kernel<6>), grid, block, 0, s, frames, w, b, gamma, beta, out, H, W, eps, (int)ntok, Hp, Wp, rf, single);
  return true;
}
#endif  // TCE_CHAIN_PART != 2
