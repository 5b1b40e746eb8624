// Shared GEMM epilogue: bias + activation + residual, specialised at compile time on (ACT, RES) so that the
// 16-element store loop of a 32x32 accumulator tile is branch-free, with the residual loads of a tile issued
// together (one wait) instead of one dependent load per element.
#pragma once
#include "common.h"

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
    if (ACT == 2) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    if (RES == 1) v += rv[r];
    if (RES == 2) v *= rv[r];
    if (row < M) C[(long long)row * ldc + col] = v;
  }
}

// runs BODY(ACT, RES) with compile-time constants chosen from the run-time (act, res_mode)
#define TCE_EPI_DISPATCH(act, res_mode, BODY) \
  switch ((act) * 3 + (res_mode)) {           \
    case 0: BODY(0, 0); break;                \
    case 1: BODY(0, 1); break;                \
    case 2: BODY(0, 2); break;                \
    case 3: BODY(1, 0); break;                \
    case 4: BODY(1, 1); break;                \
    case 5: BODY(1, 2); break;                \
    case 6: BODY(2, 0); break;                \
    case 7: BODY(2, 1); break;                \
    default: BODY(2, 2); break;               \
  }
