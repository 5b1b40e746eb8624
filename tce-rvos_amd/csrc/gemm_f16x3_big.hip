// Split-fp16 GEMM, large tiles (256x128 with 8 waves, 128x128 with 4): the dispatcher.  The kernels live in four translation
// units, one per (tile, arithmetic mode) -- gemm_f16x3_big_{256,128}_{split,single}.hip -- so that the library builds in parallel
// (every kernel carries 7 specialised epilogue bodies; see gemm_epilogue.h).
#include "common.h"
#include "../../include/tce_rvos.h"

#define TCE_BIG_PART(name)                                                     \
  int tce_gemm_f16x3_big_launch_##name(const tceGemmArgs& a, hipStream_t s);   \
  int tce_gemm_f16x3_big_stamp_##name(long long* dev_buf);                     \
  int tce_gemm_f16x3_big_epi_##name(int lds_staged);
TCE_BIG_PART(p256_split)
TCE_BIG_PART(p256_single)
TCE_BIG_PART(p128_split)
TCE_BIG_PART(p128_single)

int tce_gemm_single_pass();  // gemm.hip

int tce_gemm_f16x3_launch_big(const tceGemmArgs& a, int tile, hipStream_t s) {
  const bool single = tce_gemm_single_pass() != 0;
  if (tile == 256128) return single ? tce_gemm_f16x3_big_launch_p256_single(a, s) : tce_gemm_f16x3_big_launch_p256_split(a, s);
  return single ? tce_gemm_f16x3_big_launch_p128_single(a, s) : tce_gemm_f16x3_big_launch_p128_split(a, s);
}
int tce_gemm_f16x3_big_set_stamp(long long* dev_buf) {
  return tce_gemm_f16x3_big_stamp_p256_split(dev_buf) | tce_gemm_f16x3_big_stamp_p256_single(dev_buf) |
         tce_gemm_f16x3_big_stamp_p128_split(dev_buf) | tce_gemm_f16x3_big_stamp_p128_single(dev_buf);
}
int tce_gemm_f16x3_big_set_epilogue(int lds_staged) {
  return tce_gemm_f16x3_big_epi_p256_split(lds_staged) | tce_gemm_f16x3_big_epi_p256_single(lds_staged) |
         tce_gemm_f16x3_big_epi_p128_split(lds_staged) | tce_gemm_f16x3_big_epi_p128_single(lds_staged);
}
