// fp32-accurate GEMM / implicit-GEMM convolution on the fp16 matrix cores ("3 x fp16 split").
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32) issues at the fp32 vector rate, 1/16 of the fp16 rate.  Here every fp32
// operand x is split on the fly into two fp16 numbers
//        hi = f16(x)                 (round to nearest, 11 significant bits)
//        lo = f16((x - hi) * 2^11)   (the next 11 bits, pre-scaled so that it never underflows)
// and a product a*b is evaluated as  hi_a*hi_b  +  2^-11 * (hi_a*lo_b + lo_a*hi_b)  with three
// v_mfma_f32_32x32x16_f16 instructions accumulating in fp32 (two accumulators: main and cross).  fp16 x fp16
// products are exact in fp32, so the only terms lost are lo*lo (2^-22 relative) and the final rounding of
// lo (2^-22): the result matches an fp32 GEMM to ~3e-7 relative per product, below fp32 accumulation noise
// at the K sizes of this path -- the parity tests run against the same tolerances as the exact-fp32 kernel.
// Inputs must lie in the fp16 range (|x| < 65504); everything on this path is normalised activations / weights.
//
// Structure: 256 threads = 4 waves (2x2) over a BM x BN tile, BK = 32.  Global fp32 tiles are loaded as
// float4 (8 threads cover one 128-byte row segment), converted in registers, and stored as four fp16 LDS planes
// per stage (A_hi, A_lo, B_hi, B_lo; rows of 32 halfs = 64 B, 16-byte chunks XOR-swizzled by (row>>2)&3 so that
// the ds_read_b128 operand fetches of a 16-lane group touch all 64 banks exactly once).  Two stages, register
// prefetch of the next K slice during the MFMAs, one barrier per slice.  Per k16 step a wave issues 8
// ds_read_b128 for 12 MFMAs.  Epilogue identical to the fp32 kernel (bias, ReLU/GELU, residual add/mul).
#include "common.h"
#include "../../include/tce_rvos.h"

namespace {

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;
constexpr float LO_SCALE = 2048.0f;
constexpr float LO_INV = 1.0f / 2048.0f;

__device__ __forceinline__ void split4(const f32x4 v, h16x4& hi, h16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const _Float16 h = (_Float16)v[j];
    hi[j] = h;
    lo[j] = (_Float16)((v[j] - (float)h) * LO_SCALE);
  }
}

// byte offset of (row, 16-byte chunk c) inside a plane with 64-byte rows
__device__ __forceinline__ int swz(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

template <int BM, int BN, bool CONV>
__global__ void __launch_bounds__(256, 2) gemm_f16x3_kernel(const tceGemmArgs p, const int tiles_m, const int tiles_n) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NA = BM / 32;  // float4 per thread for the A tile (32 rows per pass)
  constexpr int NB = BN / 32;
  constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;       // bytes
  constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int bz = blockIdx.z;

  const float* __restrict__ A = p.A + (long long)bz * p.sA;
  const float* __restrict__ A2 = p.A2 ? p.A2 + (long long)bz * p.sA2 : nullptr;
  const float* __restrict__ W = p.W + (long long)bz * p.sW;
  const float* __restrict__ bias = p.bias ? p.bias + (long long)bz * p.sBias : nullptr;
  const float* __restrict__ res = p.res ? p.res + (long long)bz * p.sRes : nullptr;
  float* __restrict__ C = p.C + (long long)bz * p.sC;

  const int kq = tid & 7;     // which float4 of the 32-wide K slice
  const int lrow = tid >> 3;  // 0..31
  long long a_off[NA];
  bool a_ok[NA];
  int c_t[NA], c_y[NA], c_x[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int gm = tm * BM + lrow + 32 * i;
    a_ok[i] = gm < p.M;
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int t = gm / hw, rem = gm - t * hw;
      c_t[i] = t;
      c_y[i] = (rem / p.Wo) * p.stride - p.pad;
      c_x[i] = (rem % p.Wo) * p.stride - p.pad;
      a_off[i] = 0;
    } else {
      a_off[i] = (long long)gm * p.lda;
      c_t[i] = c_y[i] = c_x[i] = 0;
    }
  }
  long long w_off[NB];
  bool w_ok[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int gn = tn * BN + lrow + 32 * i;
    w_ok[i] = gn < p.N;
    w_off[i] = (long long)gn * p.ldw;
  }

  f32x4 ra[NA], rb[NB];
  auto load_tiles = [&](int kt) {
    const int k0 = kt * BK + kq * 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (CONV) {
        const int tap = (kt * BK) / p.Cin;
        const int c0 = k0 - tap * p.Cin;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        const int yi = c_y[i] + ky, xi = c_x[i] + kx;
        if (a_ok[i] && yi >= 0 && yi < p.H && xi >= 0 && xi < p.Wd)
          v = *reinterpret_cast<const f32x4*>(A + (((long long)c_t[i] * p.H + yi) * p.Wd + xi) * p.Cin + c0);
      } else if (a_ok[i]) {
        v = *reinterpret_cast<const f32x4*>(A + a_off[i] + k0);
        if (A2) v += *reinterpret_cast<const f32x4*>(A2 + (long long)(tm * BM + lrow + 32 * i) * p.lda2 + k0);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (w_ok[i]) v = *reinterpret_cast<const f32x4*>(W + w_off[i] + k0);
      rb[i] = v;
    }
  };
  auto store_tiles = [&](int buf) {
    unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = lrow + 32 * i;
      const int off = swz(row, kq >> 1) + ((kq & 1) << 3);
      h16x4 hi, lo;
      split4(ra[i], hi, lo);
      *reinterpret_cast<h16x4*>(st + off) = hi;
      *reinterpret_cast<h16x4*>(st + PLANE_A + off) = lo;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = lrow + 32 * i;
      const int off = swz(row, kq >> 1) + ((kq & 1) << 3);
      h16x4 hi, lo;
      split4(rb[i], hi, lo);
      *reinterpret_cast<h16x4*>(st + 2 * PLANE_A + off) = hi;
      *reinterpret_cast<h16x4*>(st + 2 * PLANE_A + PLANE_B + off) = lo;
    }
  };

  f32x16 acc[TM][TN], acx[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.f;
        acx[i][j][r] = 0.f;
      }

  const int nk = p.K / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  const int l31 = lane & 31, lhi = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);
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
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acx[i][j], 0, 0, 0);
          acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acx[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = tn * BN + wn * WN + j * 32 + l31;
    if (col >= p.N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tm * BM + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row >= p.M) continue;
        float v = fmaf(acx[i][j][r], LO_INV, acc[i][j][r]) + bv;
        if (p.act == 1) v = fmaxf(v, 0.f);
        else if (p.act == 2) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        if (p.res_mode == 1) v += res[(long long)row * p.ldres + col];
        else if (p.res_mode == 2) v *= res[(long long)row * p.ldres + col];
        C[(long long)row * p.ldc + col] = v;
      }
    }
  }
}

template <int BM, int BN>
void launch(const tceGemmArgs& a, hipStream_t s) {
  const int tiles_m = tce_cdiv(a.M, BM), tiles_n = tce_cdiv(a.N, BN);
  dim3 grid(tiles_m * tiles_n, 1, a.batch > 0 ? a.batch : 1);
  if (a.conv)
    hipLaunchKernelGGL((gemm_f16x3_kernel<BM, BN, true>), grid, dim3(256), 0, s, a, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL((gemm_f16x3_kernel<BM, BN, false>), grid, dim3(256), 0, s, a, tiles_m, tiles_n);
}

}  // namespace

// called by tce_gemm_f32 (gemm.hip) after argument validation when the split mode is selected and K % 32 == 0
int tce_gemm_f16x3_launch(const tceGemmArgs& a, int tile, hipStream_t s) {
  if (tile == 128128) launch<128, 128>(a, s);
  else if (tile == 12864) launch<128, 64>(a, s);
  else launch<64, 64>(a, s);
  return 0;
}
