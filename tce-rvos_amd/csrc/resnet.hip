// ResNet-50 front of the backbone (BASELINE config 1).
//
// Only the two layers that are not GEMM-shaped live here: the 7x7/s2 stem convolution on 3 input channels (K = 147,
// no MFMA shape worth the staging) with FrozenBatchNorm2d (models/backbone.py:46-56, folded into the weights and a
// per-channel bias by the host) + ReLU, and the 3x3/s2 max pool.  Every bottleneck convolution (1x1, 3x3, strided
// 1x1 projection) goes through tce_gemm_f32 with conv = 1 and the BN folded the same way.
//
// Layouts: frames NCHW [T,3,H,W] (what the caller hands over); everything after the stem is channels-last
// [T*h*w, C] like the rest of the path.
#include "common.h"
#include "../../include/tce_rvos.h"

namespace {

constexpr int STEM_TH = 8, STEM_TW = 16;               // output tile per workgroup
constexpr int STEM_PH = (STEM_TH - 1) * 2 + 7;         // 21 input rows
constexpr int STEM_PW = (STEM_TW - 1) * 2 + 7;         // 37 input columns
constexpr int STEM_PWP = STEM_PW + 1;                  // padded pitch
constexpr int STEM_K = 147, STEM_C = 64;

// out[t, oy, ox, :] = relu(sum_{c,ky,kx} x[t, c, 2*oy-3+ky, 2*ox-3+kx] * w[(c*7+ky)*7+kx][:] + bias[:])
__global__ void __launch_bounds__(256) resnet_stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          const int H, const int W, const int Ho, const int Wo) {
  __shared__ float sw[STEM_K * STEM_C];
  __shared__ float sp[3 * STEM_PH * STEM_PWP];
  const int tid = threadIdx.x;
  const int t = blockIdx.z, oy0 = blockIdx.y * STEM_TH, ox0 = blockIdx.x * STEM_TW;
  for (int i = tid; i < STEM_K * STEM_C / 4; i += 256)
    reinterpret_cast<f32x4*>(sw)[i] = reinterpret_cast<const f32x4*>(w)[i];
  const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
  for (int i = tid; i < 3 * STEM_PH * STEM_PW; i += 256) {
    const int c = i / (STEM_PH * STEM_PW), r = i % (STEM_PH * STEM_PW);
    const int py = r / STEM_PW, px = r % STEM_PW;
    const int iy = iy0 + py, ix = ix0 + px;
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[((long long)(t * 3 + c) * H + iy) * W + ix];
    sp[(c * STEM_PH + py) * STEM_PWP + px] = v;
  }
  __syncthreads();
  const int p = tid & 127, half = tid >> 7;  // a wave shares `half`: weight reads are LDS broadcasts
  const int py = p / STEM_TW, px = p % STEM_TW;
  float acc[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = 0.f;
  for (int c = 0; c < 3; ++c)
    for (int ky = 0; ky < 7; ++ky) {
      const float* prow = sp + (c * STEM_PH + py * 2 + ky) * STEM_PWP + px * 2;
      const float* wrow = sw + ((c * 7 + ky) * 7) * STEM_C + half * 32;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const float xv = prow[kx];
#pragma unroll
        for (int j4 = 0; j4 < 8; ++j4) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + kx * STEM_C + j4 * 4);
          acc[j4 * 4 + 0] = fmaf(xv, wv[0], acc[j4 * 4 + 0]);
          acc[j4 * 4 + 1] = fmaf(xv, wv[1], acc[j4 * 4 + 1]);
          acc[j4 * 4 + 2] = fmaf(xv, wv[2], acc[j4 * 4 + 2]);
          acc[j4 * 4 + 3] = fmaf(xv, wv[3], acc[j4 * 4 + 3]);
        }
      }
    }
  const int oy = oy0 + py, ox = ox0 + px;
  if (oy >= Ho || ox >= Wo) return;
  float* o = out + ((long long)(t * Ho + oy) * Wo + ox) * STEM_C + half * 32;
#pragma unroll
  for (int j4 = 0; j4 < 8; ++j4) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + half * 32 + j4 * 4);
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = fmaxf(acc[j4 * 4 + c] + bv[c], 0.f);
    *reinterpret_cast<f32x4*>(o + j4 * 4) = v;
  }
}

// nn.MaxPool2d(kernel 3, stride 2, padding 1): the padding never wins (-inf), channels-last, one float4 per thread
__global__ void __launch_bounds__(256) maxpool3x3s2_cl_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                              const int H, const int W, const int C4, const int Ho,
                                                              const int Wo, const long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c4 = (int)(i % C4);
  long long r = i / C4;
  const int ox = (int)(r % Wo);
  r /= Wo;
  const int oy = (int)(r % Ho);
  const int t = (int)(r / Ho);
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = oy * 2 - 1 + dy;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = ox * 2 - 1 + dx;
      if (ix < 0 || ix >= W) continue;
      const f32x4 v = reinterpret_cast<const f32x4*>(x)[((long long)(t * H + iy) * W + ix) * C4 + c4];
#pragma unroll
      for (int c = 0; c < 4; ++c) m[c] = fmaxf(m[c], v[c]);
    }
  }
  reinterpret_cast<f32x4*>(out)[i] = m;
}

}  // namespace

extern "C" int tce_resnet_stem_f32(const float* frames, const float* w_k64, const float* bias, float* out, int32_t T,
                                   int32_t H, int32_t W, tceStream stream) {
  TCE_CHECK_ARG(frames && w_k64 && bias && out, "tce_resnet_stem_f32: null pointer");
  TCE_CHECK_ARG(T > 0 && H > 0 && W > 0 && T <= 65535, "tce_resnet_stem_f32: bad shape T=%d H=%d W=%d", T, H, W);
  TCE_CHECK_ARG(tce_aligned16(w_k64) && tce_aligned16(bias) && tce_aligned16(out),
                "tce_resnet_stem_f32: w, bias and out must be 16-byte aligned");
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  dim3 grid(tce_cdiv(Wo, STEM_TW), tce_cdiv(Ho, STEM_TH), T);
  TCE_CHECK_ARG(grid.y <= 65535, "tce_resnet_stem_f32: frame too tall");
  hipLaunchKernelGGL(resnet_stem_kernel, grid, dim3(256), 0, (hipStream_t)stream, frames, w_k64, bias, out, H, W, Ho, Wo);
  TCE_CHECK_LAUNCH("tce_resnet_stem_f32");
  return TCE_OK;
}

extern "C" int tce_maxpool3x3s2_cl_f32(const float* x, float* out, int32_t T, int32_t H, int32_t W, int32_t C,
                                       tceStream stream) {
  TCE_CHECK_ARG(x && out, "tce_maxpool3x3s2_cl_f32: null pointer");
  TCE_CHECK_ARG(T > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "tce_maxpool3x3s2_cl_f32: bad shape (C=%d must be a multiple of 4)", C);
  TCE_CHECK_ARG(tce_aligned16(x) && tce_aligned16(out), "tce_maxpool3x3s2_cl_f32: x and out must be 16-byte aligned");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)T * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_cl_kernel, dim3(tce_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, H, W,
                     C / 4, Ho, Wo, total);
  TCE_CHECK_LAUNCH("tce_maxpool3x3s2_cl_f32");
  return TCE_OK;
}
