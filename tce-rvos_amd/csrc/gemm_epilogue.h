// Shared GEMM epilogue: bias + activation + residual, specialised at compile time on (ACT, RES) so that the
// 16-element store loop of a 32x32 accumulator tile is branch-free, with the residual loads of a tile issued
// together (one wait) instead of one dependent load per element.
#pragma once
#include "common.h"

// Range guard of the split-fp16 arithmetic (include/tce_rvos.h, tce_set_range_flag): running maximum over what an
// epilogue stores, kept on the BIT PATTERNS of |v| (unsigned compare is monotonic for non-negative floats and ranks
// Inf / NaN above every finite value, so they trip the limit too): 3 VALU per 2 values.
typedef unsigned tce_amax_t;
__device__ __forceinline__ unsigned tce_absbits(const float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; }
__device__ __forceinline__ tce_amax_t tce_amax4(tce_amax_t amax, const f32x4 o) {
  amax = max(amax, max(tce_absbits(o[0]), tce_absbits(o[1])));
  return max(amax, max(tce_absbits(o[2]), tce_absbits(o[3])));
}
__device__ __forceinline__ tce_amax_t tce_amax1(tce_amax_t amax, const float v) { return max(amax, tce_absbits(v)); }
__device__ __forceinline__ void tce_range_report(int* flag, const tce_amax_t amax) {
  if (flag && amax >= __builtin_bit_cast(unsigned, TCE_RANGE_LIMIT)) *flag = 1;
}

template <int ACT, int RES>
__device__ __forceinline__ void tce_epi_store(const f32x16& x, const float bv, const float* __restrict__ res,
                                              float* __restrict__ C, const int row0, const int col, const int M,
                                              const long long ldc, const long long ldres) {
  // accumulator register r holds row row0 + (r&3) + 8*(r>>2) of column `col` (row0 already includes 4*(lane>>5))
  float rv[16];
  if (RES != 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + (r & 3) + 8 * (r >> 2);
      rv[r] = res[(long long)min(row, M - 1) * ldres + col];
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + (r & 3) + 8 * (r >> 2);
    float v = x[r] + bv;
    if (ACT == 1) v = fmaxf(v, 0.f);
    if (ACT == 2) v = tce_gelu(v);
    if (RES == 1) v += rv[r];
    if (RES == 2) v *= rv[r];
    if (ACT == 3) v = fmaxf(v, 0.f);  // ReLU after the residual
    if (row < M) C[(long long)row * ldc + col] = v;
  }
}

// Transposed-accumulator form (the GEMM kernels issue mfma(W_frag, A_frag): D[i = n][j = m]): the lane owns output
// ROW m and accumulator registers 4g..4g+3 are 4 CONSECUTIVE columns n0 + 8g + (0..3)  (n0 already includes
// 4*(lane>>5)), so a 32x32 tile is written with four 16-byte stores per lane instead of sixteen 4-byte ones
// (dword-per-lane epilogues are store-issue bound at ~7 B/clk/CU on gfx950: this cut the epilogue of a
// 256x128 tile from 17k to a few thousand cycles).  vec_ok = row pitches / bases allow aligned float4 access.
template <int ACT, int RES>
__device__ __forceinline__ void tce_epi_store_t(const f32x16& x, const float* __restrict__ bias,
                                                const float* __restrict__ res, float* __restrict__ C, const int row,
                                                const int n0, const int M, const int N, const long long ldc,
                                                const long long ldres, const bool vec_ok, tce_amax_t& amax) {
  if (row >= M) return;
  float* crow = C + (long long)row * ldc;
  const float* rrow = (RES != 0) ? res + (long long)row * ldres : nullptr;
  if (vec_ok && n0 + 27 < N) {
    f32x4 rv[4], bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (RES != 0) rv[g] = *reinterpret_cast<const f32x4*>(rrow + n0 + 8 * g);
      if (bias) bv[g] = *reinterpret_cast<const f32x4*>(bias + n0 + 8 * g);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = x[4 * g + c] + (bias ? bv[g][c] : 0.f);
        if (ACT == 1) v = fmaxf(v, 0.f);
        if (ACT == 2) v = tce_gelu(v);
        if (RES == 1) v += rv[g][c];
        if (RES == 2) v *= rv[g][c];
        if (ACT == 3) v = fmaxf(v, 0.f);  // ReLU after the residual
        o[c] = v;
      }
      amax = tce_amax4(amax, o);
      *reinterpret_cast<f32x4*>(crow + n0 + 8 * g) = o;
    }
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int n = n0 + 8 * g + c;
        if (n < N) {
          float v = x[4 * g + c] + (bias ? bias[n] : 0.f);
          if (ACT == 1) v = fmaxf(v, 0.f);
          if (ACT == 2) v = tce_gelu(v);
          if (RES == 1) v += rrow[n];
          if (RES == 2) v *= rrow[n];
          if (ACT == 3) v = fmaxf(v, 0.f);  // ReLU after the residual
          amax = tce_amax1(amax, v);
          crow[n] = v;
        }
      }
  }
}

// LDS-staged form of the same epilogue: the wave parks its 32x32 accumulator tile (row-per-lane layout) in a private
// 32 x 36-float LDS buffer and reads it back with 8 lanes per row, so every global store / residual load instruction
// covers 8 FULL 128-byte lines (8 rows x 128 B) instead of 32 half-lines.  One wave's LDS operations execute in
// order, so no barrier is needed between its write and its read-back.
constexpr int TCE_EPI_LDS_FLOATS = 32 * 36;

template <int ACT, int RES>
__device__ __forceinline__ void tce_epi_store_lds(const f32x16& x, float* __restrict__ wbuf,
                                                  const float* __restrict__ bias, const float* __restrict__ res,
                                                  float* __restrict__ C, const int row0, const int col0, const int M,
                                                  const int N, const long long ldc, const long long ldres,
                                                  const bool vec_ok, const int lane, tce_amax_t& amax) {
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 v = {x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
    *reinterpret_cast<f32x4*>(&wbuf[l31 * 36 + 8 * g + 4 * lhi]) = v;
  }
  __builtin_amdgcn_wave_barrier();
  const int c4 = (lane & 7) * 4;
  const int n = col0 + c4;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  const bool full = vec_ok && n + 3 < N;
  if (bias) {
    if (full) bv = *reinterpret_cast<const f32x4*>(bias + n);
    else {
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[c] = (n + c < N) ? bias[n + c] : 0.f;
    }
  }
  f32x4 xv[4], rv[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = it * 8 + (lane >> 3);
    xv[it] = *reinterpret_cast<const f32x4*>(&wbuf[r * 36 + c4]);
    if (RES != 0) {
      const int row = min(row0 + r, M - 1);
      if (full) rv[it] = *reinterpret_cast<const f32x4*>(res + (long long)row * ldres + n);
      else {
#pragma unroll
        for (int c = 0; c < 4; ++c) rv[it][c] = (n + c < N) ? res[(long long)row * ldres + n + c] : 0.f;
      }
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = row0 + it * 8 + (lane >> 3);
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = xv[it][c] + bv[c];
      if (ACT == 1) v = fmaxf(v, 0.f);
      if (ACT == 2) v = tce_gelu(v);
      if (RES == 1) v += rv[it][c];
      if (RES == 2) v *= rv[it][c];
      if (ACT == 3) v = fmaxf(v, 0.f);  // ReLU after the residual
      o[c] = v;
    }
    amax = tce_amax4(amax, o);
    if (row < M) {
      if (full) *reinterpret_cast<f32x4*>(C + (long long)row * ldc + n) = o;
      else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (n + c < N) C[(long long)row * ldc + n + c] = o[c];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bool tce_epi_vec_ok(const float* C, long long ldc, const float* res, long long ldres,
                                               const float* bias, int res_mode) {
  bool ok = ((ldc & 3) == 0) && ((((uintptr_t)C) & 15u) == 0) && (!bias || ((((uintptr_t)bias) & 15u) == 0));
  if (res_mode != 0) ok = ok && ((ldres & 3) == 0) && ((((uintptr_t)res) & 15u) == 0);
  return ok;
}

// runs BODY(ACT, RES) with compile-time constants chosen from the run-time (act, res_mode).  Only the combinations the
// path uses are specialised (every one is another inlined copy of the epilogue in every GEMM kernel, and compile time
// is proportional): none, +res, *res, ReLU, ReLU then +res, GELU, ReLU after +res.  The host (tce_gemm_f32) folds
// every other legal combination onto these plus one elementwise pass (tce_epi_supported / epi_fix_kernel, gemm.hip).
#define TCE_EPI_DISPATCH(act, res_mode, BODY) \
  switch ((act) * 3 + (res_mode)) {           \
    case 0: BODY(0, 0); break;                \
    case 1: BODY(0, 1); break;                \
    case 2: BODY(0, 2); break;                \
    case 3: BODY(1, 0); break;                \
    case 4: BODY(1, 1); break;                \
    case 6: BODY(2, 0); break;                \
    default: BODY(3, 1); break;               \
  }

static inline bool tce_epi_supported(int act, int res_mode) {
  const int c = act * 3 + res_mode;
  return c == 0 || c == 1 || c == 2 || c == 3 || c == 4 || c == 6 || c == 10;
}
