// fp32-accurate GEMM / implicit-GEMM convolution on the fp16 matrix cores ("3 x fp16 split").
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32) issues at the fp32 vector rate, 1/16 of the fp16 rate.  Here every fp32
// operand x is split on the fly into two fp16 numbers
//        hi = f16(x)        (truncated, 11 significant bits)
//        lo = f16(x - hi)   (the exact remainder, truncated: the next 10-11 bits)
// and a product a*b is evaluated as  hi_a*hi_b + hi_a*lo_b + lo_a*hi_b  with three v_mfma_f32_32x32x16_f16
// instructions accumulating into ONE fp32 accumulator.  fp16 x fp16 products are exact in fp32, so the only terms
// lost are lo*lo and the truncation of lo: |x - hi - lo| <= max(2^-20 |x|, 6e-8) (the absolute floor is the fp16
// subnormal spacing, reached for |x| < 0.06).  Measured against fp64 the result is as accurate as the exact fp32
// MFMA kernel (tests/test_kernels_gpu.py::test_gemm_split_fp16_is_fp32_accurate) and the end-to-end parity tests
// run at the same tolerances in both modes.  Inputs must lie in the fp16 range (|x| < 65504); everything on this
// path is normalised activations / weights.  (An earlier variant pre-scaled lo by 2^11 into a second accumulator:
// slightly better for tiny operands, but the 64 extra accumulator registers left no room to keep more than one
// K slice of loads in flight, and the K loop was memory-latency bound.)
//
// Structure: 256 threads = 4 waves (2x2) over a BM x BN tile, BK = 32.  Global fp32 tiles are loaded as
// float4 (8 threads cover one 128-byte row segment), converted in registers, and stored as four fp16 LDS planes
// per stage (A_hi, A_lo, B_hi, B_lo; rows of 32 halfs = 64 B, 16-byte chunks XOR-swizzled by (row>>2)&3 so that
// the ds_read_b128 operand fetches of a 16-lane group touch all 64 banks exactly once).  Two stages, register
// prefetch of the next K slice during the MFMAs, one barrier per slice.  Per k16 step a wave issues 8
// ds_read_b128 for 12 MFMAs.  Epilogue identical to the fp32 kernel (bias, ReLU/GELU, residual add/mul).
#pragma once
#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/tce_rvos.h"

namespace {

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;

__device__ __forceinline__ void split4(const f32x4 v, h16x4& hi, h16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const _Float16 h = (_Float16)v[j];
    hi[j] = h;
    lo[j] = (_Float16)(v[j] - (float)h);
  }
}

// byte offset of (row, 16-byte chunk c) inside a plane with 64-byte rows
__device__ __forceinline__ int swz(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
struct h4pair {
  fp16x2_t a, b;
};

// hi by truncation (v_cvt_pkrtz_f16_f32), lo = the exact remainder, truncated.  single (tce_set_gemm_mode(2)): hi is
// the value rounded to NEAREST fp16 and lo is not used
__device__ __forceinline__ void split4_rtz(const f32x4 v, h4pair& hi, h4pair& lo, const int single = 0) {
  if (single) {
    hi.a = fp16x2_t{(__fp16)v[0], (__fp16)v[1]};
    hi.b = fp16x2_t{(__fp16)v[2], (__fp16)v[3]};
    lo.a = lo.b = fp16x2_t{(__fp16)0.f, (__fp16)0.f};
    return;
  }
  hi.a = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
  hi.b = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
  const float d0 = v[0] - (float)hi.a[0], d1 = v[1] - (float)hi.a[1];
  const float d2 = v[2] - (float)hi.b[0], d3 = v[3] - (float)hi.b[1];
  lo.a = __builtin_amdgcn_cvt_pkrtz(d0, d1);
  lo.b = __builtin_amdgcn_cvt_pkrtz(d2, d3);
}

// Diagnostic stamps (tce_debug_set_stamp_buffer): when a buffer is registered, lane 0 of wave 0 of the first 2048
// workgroups of the symmetric kernel records s_memtime at entry / after the prologue / after the K loop / after
// the epilogue stores were issued / after they drained.  NULL (the default) costs one uniform branch.
static __device__ long long* g_stamp_buf = nullptr;
static __device__ int g_epi_lds = 1;  // tuning aid: 1 = LDS-staged coalesced epilogue, 0 = direct row-per-lane stores

// Symmetric kernel: every wave loads, converts and multiplies.  DEPTH K-slices are kept in flight per thread in a
// register ring (loads are unconditional -- clamped addresses, validity applied at commit -- and the steady-state
// loop is branch-free, so the compiler emits exact counted vmcnt waits): a K step no longer pays a full memory
// round trip, which is what bounds the small / short-K problems of this path.
// SINGLE (tce_set_gemm_mode(2): one MFMA per product) is a compile-time parameter: as a run-time flag it put a uniform branch
// in front of every tile's two lo-term MFMAs and around every hi / lo split (see ffn_fused_kernel in chain.hip).
template <int BM, int BN, int WAVES_M, bool CONV, bool HAS_A2, int DEPTH, bool SINGLE>
__global__ void __launch_bounds__(128 * WAVES_M, WAVES_M == 2 ? 2 : 1)
    gemm_f16x3_kernel(const tceGemmArgs p, const int tiles_m, const int tiles_n, int* const range_flag) {
  constexpr int single = SINGLE;
  constexpr int NT = 128 * WAVES_M;                   // threads: WAVES_M x 2 waves
  constexpr int RP = NT / 8;                          // tile rows covered by one loader pass
  constexpr int WM = BM / WAVES_M, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NA = BM / RP;  // float4 per thread for the A tile
  constexpr int NB = BN / RP;
  constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;  // bytes
  constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int bz = blockIdx.z;
  long long* const stamps = (g_stamp_buf && blockIdx.x < 2048 && tid == 0) ? g_stamp_buf + blockIdx.x * 8 : nullptr;
  if (stamps) stamps[0] = (long long)__builtin_amdgcn_s_memtime();

  const float* __restrict__ A = p.A + (long long)bz * p.sA;
  const float* __restrict__ A2 = HAS_A2 ? p.A2 + (long long)bz * p.sA2 : nullptr;
  const float* __restrict__ W = p.W + (long long)bz * p.sW;
  const float* __restrict__ bias = p.bias ? p.bias + (long long)bz * p.sBias : nullptr;
  const float* __restrict__ res = p.res ? p.res + (long long)bz * p.sRes : nullptr;
  float* __restrict__ C = p.C + (long long)bz * p.sC;

  const int kq = tid & 7;     // which float4 of the 32-wide K slice
  const int lrow = tid >> 3;  // 0..RP-1
  const float* pa[NA];
  const float* pa2[NA];
  unsigned rowmask = 0;  // bit i: A row i inside M; bit 8+i: W row i inside N
  int c_y[NA], c_x[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int gm = tm * BM + lrow + RP * i;
    if (gm < p.M) rowmask |= 1u << i;
    const int gmc = min(gm, p.M - 1);
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int t = gmc / hw, rem = gmc - t * hw;
      c_y[i] = (rem / p.Wo) * p.stride - p.pad;
      c_x[i] = (rem % p.Wo) * p.stride - p.pad;
      pa[i] = A + (long long)t * p.H * p.Wd * p.Cin;
      pa2[i] = nullptr;
    } else {
      pa[i] = A + (long long)gmc * p.lda;
      pa2[i] = HAS_A2 ? A2 + (long long)gmc * p.lda2 : nullptr;
      c_y[i] = c_x[i] = 0;
    }
  }
  const float* pw[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int gn = tn * BN + lrow + RP * i;
    if (gn < p.N) rowmask |= 1u << (8 + i);
    pw[i] = W + (long long)min(gn, p.N - 1) * p.ldw;
  }
  const int nk = p.K / BK;

  f32x4 ra[DEPTH][NA], rb[DEPTH][NB], ra2[HAS_A2 ? DEPTH : 1][NA];
  unsigned okm[DEPTH];
  auto issue = [&](f32x4 (&xa)[NA], f32x4 (&xb)[NB], f32x4 (&xa2)[NA], unsigned& mask, int kt) {
    kt = min(kt, nk - 1);  // past the end: re-read the last slice (never committed)
    const int k0 = kt * BK + kq * 4;
    unsigned mk = rowmask;
    int ky = 0, kx = 0, c0 = k0;
    if (CONV) {
      // a batched convolution launch is a split-K launch (tce_gemm_splitk_f32): batch index = K chunk, so the tap /
      // channel of a slice come from its GLOBAL slice number; A is not offset, W and C are (sW, sC)
      const int ktg = kt + bz * nk;
      const int tap = (ktg * BK) / p.Cin;
      c0 = ktg * BK + kq * 4 - tap * p.Cin;
      ky = tap / p.kw;
      kx = tap - ky * p.kw;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (CONV) {
        const int yi = c_y[i] + ky, xi = c_x[i] + kx;
        if (!(yi >= 0 && yi < p.H && xi >= 0 && xi < p.Wd)) mk &= ~(1u << i);
        const int yc = min(max(yi, 0), p.H - 1), xc = min(max(xi, 0), p.Wd - 1);
        xa[i] = *reinterpret_cast<const f32x4*>(pa[i] + ((long long)yc * p.Wd + xc) * p.Cin + c0);
      } else {
        xa[i] = *reinterpret_cast<const f32x4*>(pa[i] + k0);
        if (HAS_A2) xa2[i] = *reinterpret_cast<const f32x4*>(pa2[i] + k0);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) xb[i] = *reinterpret_cast<const f32x4*>(pw[i] + k0);
    mask = mk;
  };
  auto commit = [&](const f32x4 (&xa)[NA], const f32x4 (&xb)[NB], const f32x4 (&xa2)[NA], unsigned mask, int buf) {
    unsigned char* st = smem + buf * STAGE;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = lrow + RP * i;
      const int off = swz(row, kq >> 1) + ((kq & 1) << 3);
      f32x4 av = xa[i];
      if (HAS_A2) av += xa2[i];
      h4pair hi, lo;
      split4_rtz(((mask >> i) & 1u) ? av : zero, hi, lo, single);
      *reinterpret_cast<h4pair*>(st + off) = hi;
      *reinterpret_cast<h4pair*>(st + PLANE_A + off) = lo;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = lrow + RP * i;
      const int off = swz(row, kq >> 1) + ((kq & 1) << 3);
      h4pair hi, lo;
      split4_rtz(((mask >> (8 + i)) & 1u) ? xb[i] : zero, hi, lo, single);
      *reinterpret_cast<h4pair*>(st + 2 * PLANE_A + off) = hi;
      *reinterpret_cast<h4pair*>(st + 2 * PLANE_A + PLANE_B + off) = lo;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int l31 = lane & 31, lhi = lane >> 5;
  auto compute = [&](int buf) {
    const unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 2 + lhi;
      h16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int off = swz(wm * WM + i * 32 + l31, c);
        ah[i] = *reinterpret_cast<const h16x8*>(st + off);
        al[i] = *reinterpret_cast<const h16x8*>(st + PLANE_A + off);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int off = swz(wn * WN + j * 32 + l31, c);
        bh[j] = *reinterpret_cast<const h16x8*>(st + 2 * PLANE_A + off);
        bl[j] = *reinterpret_cast<const h16x8*>(st + 2 * PLANE_A + PLANE_B + off);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // operands swapped: D[n][m] -- the lane owns an output row (see tce_epi_store_t)
          if (!single) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        }
    }
  };

  // prologue: slices 0..DEPTH-1 in flight; slice 0 committed; slice DEPTH issued into the freed slot
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(ra[d], rb[d], ra2[HAS_A2 ? d : 0], okm[d], d);
  commit(ra[0], rb[0], ra2[0], okm[0], 0);
  issue(ra[0], rb[0], ra2[0], okm[0], DEPTH);
  __syncthreads();
  if (stamps) stamps[1] = (long long)__builtin_amdgcn_s_memtime();
  int kt0 = 0;
  for (; kt0 + DEPTH <= nk; kt0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int kt = kt0 + d;
      const int r = (d + 1) % DEPTH;  // ring slot holding slice kt+1
      compute(kt & 1);
      commit(ra[r], rb[r], ra2[HAS_A2 ? r : 0], okm[r], (kt + 1) & 1);
      issue(ra[r], rb[r], ra2[HAS_A2 ? r : 0], okm[r], kt + 1 + DEPTH);
      __syncthreads();
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int kt = kt0 + d;
    if (kt < nk) {
      const int r = (d + 1) % DEPTH;
      compute(kt & 1);
      commit(ra[r], rb[r], ra2[HAS_A2 ? r : 0], okm[r], (kt + 1) & 1);
      __syncthreads();
    }
  }
  if (stamps) stamps[2] = (long long)__builtin_amdgcn_s_memtime();

  const bool vec_ok = tce_epi_vec_ok(C, p.ldc, res, p.ldres, bias, p.res_mode);
  // the K loop ended on a workgroup barrier: the stage buffers are free, each wave takes a private slice
  static_assert(2 * STAGE >= (NT / 64) * TCE_EPI_LDS_FLOATS * 4, "stage LDS too small for the epilogue buffers");
  float* wbuf = reinterpret_cast<float*>(smem) + wave * TCE_EPI_LDS_FLOATS;
  // measured (tools/gemm_stamps.py): LDS staging pays for the 8-wave 256x128 tile (9.1k -> 7.7k cycles), not for the
  // 4-wave tiles (3.6k -> 5.0k), where direct row-per-lane float4 stores stay
  const bool epi_lds = (WAVES_M == 4) && g_epi_lds != 0;
  tce_amax_t amax = 0;
#define EPI_BODY(ACT, RES)                                                                                  \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                          \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                        \
      if (epi_lds)                                                                                          \
        tce_epi_store_lds<ACT, RES>(acc[i][j], wbuf, bias, res, C, tm * BM + wm * WM + i * 32,              \
                                    tn * BN + wn * WN + j * 32, p.M, p.N, p.ldc, p.ldres, vec_ok, lane,     \
                                    amax);                                                                  \
      else                                                                                                  \
        tce_epi_store_t<ACT, RES>(acc[i][j], bias, res, C, tm * BM + wm * WM + i * 32 + l31,                \
                                  tn * BN + wn * WN + j * 32 + 4 * lhi, p.M, p.N, p.ldc, p.ldres, vec_ok,   \
                                  amax);                                                                    \
    }                                                                                                       \
  }
  TCE_EPI_DISPATCH(p.act, p.res_mode, EPI_BODY)
#undef EPI_BODY
  tce_range_report(range_flag, amax);
  if (stamps) {
    stamps[3] = (long long)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamps[4] = (long long)__builtin_amdgcn_s_memtime();
    stamps[5] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}

// the three operand forms (implicit-GEMM convolution, second addend, plain) of one tile in ONE arithmetic mode
template <int BM, int BN, int WAVES_M, int DEPTH, bool SINGLE>
void launch_mode(const tceGemmArgs& a, hipStream_t s) {
  const int tiles_m = tce_cdiv(a.M, BM), tiles_n = tce_cdiv(a.N, BN);
  dim3 grid(tiles_m * tiles_n, 1, a.batch > 0 ? a.batch : 1), block(128 * WAVES_M);
  int* const rf = tce_range_flag();
  if (a.conv) hipLaunchKernelGGL((gemm_f16x3_kernel<BM, BN, WAVES_M, true, false, DEPTH, SINGLE>), grid, block, 0, s, a, tiles_m, tiles_n, rf);
  else if (a.A2) hipLaunchKernelGGL((gemm_f16x3_kernel<BM, BN, WAVES_M, false, true, DEPTH, SINGLE>), grid, block, 0, s, a, tiles_m, tiles_n, rf);
  else hipLaunchKernelGGL((gemm_f16x3_kernel<BM, BN, WAVES_M, false, false, DEPTH, SINGLE>), grid, block, 0, s, a, tiles_m, tiles_n, rf);
}

template <int BM, int BN, int WAVES_M, int DEPTH>
void launch(const tceGemmArgs& a, hipStream_t s) {
  if (tce_gemm_single_pass()) launch_mode<BM, BN, WAVES_M, DEPTH, true>(a, s);
  else launch_mode<BM, BN, WAVES_M, DEPTH, false>(a, s);
}


inline int set_stamp_buffer(long long* dev_buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : -1;
}
inline int set_epilogue_mode(int lds_staged) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_epi_lds), &lds_staged, sizeof(lds_staged)) == hipSuccess ? 0 : -1;
}

}  // namespace
