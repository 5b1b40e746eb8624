// Split-fp16 GEMM, small-tile instantiations (128x64 and 64x64, 4 waves) + the diagnostics entry points.
#include "gemm_f16x3_kernel.h"

int tce_gemm_f16x3_big_set_stamp(long long* dev_buf);
int tce_gemm_f16x3_big_set_epilogue(int lds_staged);

int tce_gemm_f16x3_launch_small(const tceGemmArgs& a, int tile, hipStream_t s) {
  if (tile == 12864) {
    // (A three-slice register ring for this tile at low occupancy -- Swin-T stage 3's fc2, 216 workgroups walking 48 K slices --
    // was built and measured in round 5: 4600 x 384 x 1536 36.7 us with one slice in flight, 40.2 us with three; the other shapes
    // within 1 us, the clip 6.03 vs 6.07 ms.  Not kept: the tile is not bound by its loads' latency.)
    launch<128, 64, 2, 1>(a, s);
  } else {
    // few workgroups = nothing else on the CU to hide a K step's memory round trip: keep 4 slices in flight
    const long long blocks = (long long)tce_cdiv(a.M, 64) * tce_cdiv(a.N, 64) * (a.batch > 0 ? a.batch : 1);
    if (blocks < 512 || tile == 6465) launch<64, 64, 2, 4>(a, s);
    else launch<64, 64, 2, 1>(a, s);
  }
  return 0;
}

extern "C" int tce_debug_set_epilogue(int32_t lds_staged) {
  if (set_epilogue_mode(lds_staged) != 0 || tce_gemm_f16x3_big_set_epilogue(lds_staged) != 0) {
    tce_set_error("tce_debug_set_epilogue: hipMemcpyToSymbol failed");
    return TCE_ELAUNCH;
  }
  return TCE_OK;
}

extern "C" int tce_debug_set_stamp_buffer(long long* dev_buf) {
  if (set_stamp_buffer(dev_buf) != 0 || tce_gemm_f16x3_big_set_stamp(dev_buf) != 0) {
    tce_set_error("tce_debug_set_stamp_buffer: hipMemcpyToSymbol failed");
    return TCE_ELAUNCH;
  }
  return TCE_OK;
}
