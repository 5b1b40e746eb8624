// Clip front-end (inference_ytvos.py:38-42,284-287 / inference_davis.py transform):
//   T.Resize(360) on the decoded RGB frame -> T.ToTensor() -> T.Normalize(mean, std)
// torchvision's Resize on a PIL image is Pillow's Image.resize(BILINEAR): a separable, antialiased triangle filter
// in 8-bit fixed point (coefficients rounded to 22 fractional bits, +0.5 ulp, each pass clipped back to uint8),
// horizontal pass first.  The two kernels below restate exactly that arithmetic (integer, so results are
// bit-identical to Pillow's); the host computes the coefficient tables in double precision the way Pillow's
// precompute_coeffs does (tce_rvos_amd/frontend.py).  ToTensor + Normalize are fused into the second pass through a
// 3x256 table of ((v/255) - mean[c]) / std[c] evaluated in fp32 by the host, so the floats are bit-identical too.
// Both passes are HBM-streaming byte kernels: horizontal reads each input row once (taps overlap in L1/L2),
// vertical reads ksize rows per output row with coalesced byte columns.
#include "common.h"
#include "../../include/tce_rvos.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// in [T*H, Win, 3] u8 -> tmp [T*H, Wout, 3] u8; one thread per (row, x_out), 3 channels
__global__ void __launch_bounds__(256) resize_h_kernel(const uint8_t* __restrict__ in, const int* __restrict__ coef,
                                                       const int* __restrict__ bounds, uint8_t* __restrict__ tmp,
                                                       const long long rows, const int Win, const int Wout,
                                                       const int ksize) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * Wout) return;
  const int xo = (int)(i % Wout);
  const long long row = i / Wout;
  const int xmin = bounds[2 * xo], n = bounds[2 * xo + 1];
  const int* k = coef + (long long)xo * ksize;
  const uint8_t* src = in + (row * Win + xmin) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < n; ++x) {
    const int kv = k[x];
    s0 += (int)src[3 * x + 0] * kv;
    s1 += (int)src[3 * x + 1] * kv;
    s2 += (int)src[3 * x + 2] * kv;
  }
  uint8_t* d = tmp + i * 3;
  d[0] = (uint8_t)clip8(s0);
  d[1] = (uint8_t)clip8(s1);
  d[2] = (uint8_t)clip8(s2);
}

// tmp [T, Hin, W, 3] u8 -> out [T, 3, Hout, W] f32 through lut[3][256]; one thread per (t, y_out, x)
__global__ void __launch_bounds__(256) resize_v_norm_kernel(const uint8_t* __restrict__ tmp, const int* __restrict__ coef,
                                                            const int* __restrict__ bounds, const float* __restrict__ lut,
                                                            float* __restrict__ out, const int T, const int Hin,
                                                            const int W, const int Hout, const int ksize) {
  __shared__ float slut[768];
  for (int j = threadIdx.x; j < 768; j += 256) slut[j] = lut[j];
  __syncthreads();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)T * Hout * W) return;
  const int x = (int)(i % W);
  const int yo = (int)((i / W) % Hout);
  const int t = (int)(i / ((long long)W * Hout));
  const int ymin = bounds[2 * yo], n = bounds[2 * yo + 1];
  const int* k = coef + (long long)yo * ksize;
  const uint8_t* src = tmp + (((long long)t * Hin + ymin) * W + x) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < n; ++y) {
    const int kv = k[y];
    const uint8_t* p = src + (long long)y * W * 3;
    s0 += (int)p[0] * kv;
    s1 += (int)p[1] * kv;
    s2 += (int)p[2] * kv;
  }
  const long long plane = (long long)Hout * W;
  float* o = out + (long long)t * 3 * plane + (long long)yo * W + x;
  o[0] = slut[clip8(s0)];
  o[plane] = slut[256 + clip8(s1)];
  o[2 * plane] = slut[512 + clip8(s2)];
}

}  // namespace

extern "C" int tce_resize_h_u8(const uint8_t* in, const int32_t* coef, const int32_t* bounds, uint8_t* tmp, int64_t rows,
                               int32_t Win, int32_t Wout, int32_t ksize, tceStream stream) {
  TCE_CHECK_ARG(in && coef && bounds && tmp, "tce_resize_h_u8: null pointer");
  TCE_CHECK_ARG(rows > 0 && Win > 0 && Wout > 0 && ksize > 0, "tce_resize_h_u8: bad shape");
  TCE_CHECK_ARG(ksize <= 2 * Win + 3, "tce_resize_h_u8: ksize=%d is inconsistent with Win=%d", ksize, Win);
  const long long total = rows * Wout;
  hipLaunchKernelGGL(resize_h_kernel, dim3(tce_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, coef, bounds, tmp,
                     (long long)rows, Win, Wout, ksize);
  TCE_CHECK_LAUNCH("tce_resize_h_u8");
  return TCE_OK;
}

extern "C" int tce_resize_v_norm_f32(const uint8_t* tmp, const int32_t* coef, const int32_t* bounds, const float* lut,
                                     float* out, int32_t T, int32_t Hin, int32_t W, int32_t Hout, int32_t ksize,
                                     tceStream stream) {
  TCE_CHECK_ARG(tmp && coef && bounds && lut && out, "tce_resize_v_norm_f32: null pointer");
  TCE_CHECK_ARG(T > 0 && Hin > 0 && W > 0 && Hout > 0 && ksize > 0, "tce_resize_v_norm_f32: bad shape");
  const long long total = (long long)T * Hout * W;
  hipLaunchKernelGGL(resize_v_norm_kernel, dim3(tce_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, tmp, coef, bounds,
                     lut, out, T, Hin, W, Hout, ksize);
  TCE_CHECK_LAUNCH("tce_resize_v_norm_f32");
  return TCE_OK;
}
