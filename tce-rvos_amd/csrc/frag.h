// Fragment-level helpers shared by the token-stationary kernels (chain.hip, swinattn.hip): fp16 hi/lo operand splitting,
// the 1 KiB "piece" = one wave-wide 16-byte-per-lane LDS-DMA / ds_read_b128, and the per-wave LDS staging tile.
#pragma once
#include "common.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PIECE = 1024;  // one wave-wide 16-byte-per-lane DMA / ds_read_b128

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

// 8 fp32 -> fp16 hi (truncated) + fp16 lo (truncated exact remainder); element j of the fragment = v[j].
// single (tce_set_gemm_mode(2)): hi = the value rounded to NEAREST fp16, lo unused (zero)
__device__ __forceinline__ HL split8(const float* v, const int single = 0) {
  u32x4 h, l;
  if (single) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      h[q] = __builtin_bit_cast(unsigned, fp16x2_t{(__fp16)v[2 * q], (__fp16)v[2 * q + 1]});
      l[q] = 0u;
    }
    HL r;
    r.hi = __builtin_bit_cast(h16x8, h);
    r.lo = __builtin_bit_cast(h16x8, l);
    return r;
  }
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

// Per-wave staging tile in LDS: 32 rows x 32 floats, 144-byte pitch (see chain.hip)
constexpr int WT_PITCH = 36;                 // floats
constexpr int WT_BYTES = 32 * WT_PITCH * 4;  // 4608 bytes per wave

}  // namespace
