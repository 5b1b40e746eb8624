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
