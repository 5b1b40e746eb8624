// Shared helpers for the gfx950 kernels.  CDNA4 only: 64-lane wavefronts, MFMA, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define TCE_OK 0
#define TCE_EINVAL (-1)
#define TCE_ELAUNCH (-2)

void tce_set_error(const char* fmt, ...);
// device int32 registered with tce_set_range_flag (NULL = check disabled); see include/tce_rvos.h
int* tce_range_flag();
#define TCE_RANGE_LIMIT 60000.f
// 1 when tce_set_gemm_mode(2) is active: the fp16 kernels issue ONE MFMA per product on operands rounded to nearest
// fp16 (fp32 accumulate) instead of the three of the hi/lo split (defined in gemm.hip)
int tce_gemm_single_pass();
bool tce_patch_embed_mfma(const float* frames, const float* w, const float* b, const float* gamma, const float* beta,
                          float* out, int H, int W, int C, float eps, long long ntok, int Hp, int Wp, hipStream_t s);

#define TCE_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      tce_set_error(__VA_ARGS__);           \
      return TCE_EINVAL;                    \
    }                                       \
  } while (0)

#define TCE_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      tce_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return TCE_ELAUNCH;                                                   \
    }                                                                       \
  } while (0)

static inline bool tce_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int tce_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// XCD-aware remap of a linear workgroup id: the dispatcher deals consecutive ids round-robin over the 8
// XCDs (ids b and b+8 share an L2), so give each XCD a contiguous chunk of the tile space.  Bijective for
// any grid size (cdna guide section 5, "XCD swizzle must be bijective").  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

#if defined(__HIPCC__)
// erf(x), BRANCH-FREE.  The device library's erff takes one of two paths per lane (|x| < 1: odd polynomial; else 1 - exp(-poly)) under
// exec masks -- inside a software-pipelined MFMA loop (GELU of the Swin MLPs) that is a pair of divergent branches per value, and a
// wave holds values of both kinds nearly always.  Same two polynomials (the library's coefficients), both evaluated, one select;
// the exponential is one v_exp_f32 of the rounded product (|error| <= 1e-7 absolute, one ulp of the result, in [0.84, 1)).
__device__ __forceinline__ float tce_erff(const float x) {
  const float ax = __builtin_fabsf(x);
  float p = __builtin_fmaf(ax, __builtin_bit_cast(float, 0x378e98abu), __builtin_bit_cast(float, 0xb9c68948u));
  p = __builtin_fmaf(ax, p, __builtin_bit_cast(float, 0x3b7cd369u));
  p = __builtin_fmaf(ax, p, __builtin_bit_cast(float, 0xbcc618b2u));
  p = __builtin_fmaf(ax, p, __builtin_bit_cast(float, 0x3dda74e4u));
  p = __builtin_fmaf(ax, p, __builtin_bit_cast(float, 0x3f228afdu));
  p = __builtin_fmaf(ax, p, __builtin_bit_cast(float, 0x3e03c728u));
  const float t = __builtin_fmaf(ax, p, ax);
  const float big = 1.0f - __builtin_amdgcn_exp2f(-1.4426950408889634f * t);
  const float s = x * x;
  float q = __builtin_fmaf(s, __builtin_bit_cast(float, 0xba1345e1u), __builtin_bit_cast(float, 0x3ba10414u));
  q = __builtin_fmaf(s, q, __builtin_bit_cast(float, 0xbcdac9b8u));
  q = __builtin_fmaf(s, q, __builtin_bit_cast(float, 0x3de703beu));
  q = __builtin_fmaf(s, q, __builtin_bit_cast(float, 0xbec09330u));
  q = __builtin_fmaf(s, q, __builtin_bit_cast(float, 0x3e0375d0u));
  const float small = __builtin_fmaf(ax, q, ax);
  return __builtin_copysignf(ax < 1.0f ? small : big, x);
}
// GELU (erf form, nn.GELU's default)
__device__ __forceinline__ float tce_gelu(const float v) { return 0.5f * v * (1.f + tce_erff(v * 0.70710678118654752440f)); }
#endif
