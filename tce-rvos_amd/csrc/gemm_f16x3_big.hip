// Split-fp16 GEMM, large-tile instantiations (256x128 with 8 waves, 128x128 with 4): one translation unit per tile
// family so the library builds in parallel (every kernel carries 7 specialised epilogue bodies; see gemm_epilogue.h).
#include "gemm_f16x3_kernel.h"

int tce_gemm_f16x3_launch_big(const tceGemmArgs& a, int tile, hipStream_t s) {
  if (tile == 256128) launch<256, 128, 4, 1>(a, s);  // 512 threads, 8 waves of 64x64
  else launch<128, 128, 2, 1>(a, s);
  return 0;
}
int tce_gemm_f16x3_big_set_stamp(long long* dev_buf) { return set_stamp_buffer(dev_buf); }
int tce_gemm_f16x3_big_set_epilogue(int lds_staged) { return set_epilogue_mode(lds_staged); }
