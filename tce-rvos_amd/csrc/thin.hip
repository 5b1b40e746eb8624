// Thin linear layers on the matrix cores: M <= a few dozen rows against a wide, deep weight matrix -- RoBERTa's dense
// layers at 32 tokens (tce_rvos.py:406-424: 768 x 2304 / 768 / 3072 and 3072 x 768), 28 MB of fp32 weights per encoder
// layer that are touched exactly once per clip.  Such a product is a weight STREAM: the only thing that matters is that
// every byte of W is requested early, by many workgroups, in one memory round trip.  The tiled GEMM's split-K form takes
// 13 us per projection here (a 64 x 64 tile half empty, a two-stage LDS pipeline of six short K slices: three dependent
// round trips); this kernel takes one:
//
//   grid (N/32, ceil(M/32), K/256); a workgroup = 32 output columns x 32 rows x 256 of K; each of its 4 waves loads ITS 64
//   of K of the W slab (32 rows x 256 B per wave, sixteen 16-byte loads per lane, all in flight at once) and of x straight
//   into registers in MFMA fragment order, splits both to fp16 hi/lo on the fly (W is static but streamed once: a
//   pre-split copy would double its bytes), multiplies (x as A, W as B: the accumulator has the output column on the lane),
//   the four waves' tiles are summed through LDS and the sums go to a partial plane ws[kz][M][N] in full 128-byte rows.
//   The partial planes (K/256 of them) meet in the reduction kernels of gemm.hip (bias, activation, residual, LayerNorm)
//   or in the NEXT consumer's load: PRO = the x operand is itself given as partial planes + bias (+ GELU) of the previous
//   layer (fc1 -> fc2 needs no reduction launch); tce_mha_small64_splits_f32 reads q, k, v the same way.
// Contraction order inside a 32-wide K pair: fragment slot (hf, j) of step 0 <-> k = 16 hf + j, of step 1 <-> k = 16 hf + 8 + j
// (both operands alike), so a lane reads 64 contiguous bytes of its row per pair.
#include "common.h"
#include "frag.h"
#include "../../include/tce_rvos.h"

namespace {

struct ThinArgs {
  const float* x;       // [M, K] (row pitch ldx), or PRO: partial planes [xsplits][M][K] (dense)
  const float* bias_x;  // PRO: [K] added to the plane sum (may be NULL)
  const float* W;       // [N, K], row pitch ldw
  float* ws;            // [K/256][M][N]
  long long ldx, ldw;
  int M, N, K, xsplits, act_x, single;
};

__device__ __forceinline__ int crow_t(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <bool PRO>
__global__ void __launch_bounds__(256) thin_partials_kernel(const ThinArgs p) {
  __shared__ __attribute__((aligned(16))) float sRed[4][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hf = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32, kz = blockIdx.z;
  const int k0 = kz * 256 + wave * 64 + 16 * hf;  // this lane's first k of pair 0 (pair 1: + 32)
  const int m = min(m0 + l31, p.M - 1);
  f32x4 wv[2][4], xv[2][4];
  const float* wr = p.W + (long long)(n0 + l31) * p.ldw + k0;
#pragma unroll
  for (int pr = 0; pr < 2; ++pr)
#pragma unroll
    for (int i = 0; i < 4; ++i) wv[pr][i] = *reinterpret_cast<const f32x4*>(wr + 32 * pr + 4 * i);
  if (!PRO) {
    const float* xr = p.x + (long long)m * p.ldx + k0;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[pr][i] = *reinterpret_cast<const f32x4*>(xr + 32 * pr + 4 * i);
  } else {
    const long long plane = (long long)p.M * p.K;
    const float* xr = p.x + (long long)m * p.K + k0;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 a = *reinterpret_cast<const f32x4*>(xr + 32 * pr + 4 * i);
        for (int s = 1; s < p.xsplits; ++s) a += *reinterpret_cast<const f32x4*>(xr + s * plane + 32 * pr + 4 * i);
        if (p.bias_x) a += *reinterpret_cast<const f32x4*>(p.bias_x + k0 + 32 * pr + 4 * i);
        if (p.act_x == 1) {
#pragma unroll
          for (int c = 0; c < 4; ++c) a[c] = fmaxf(a[c], 0.f);
        } else if (p.act_x == 2) {
#pragma unroll
          for (int c = 0; c < 4; ++c) a[c] = 0.5f * a[c] * (1.f + erff(a[c] * 0.70710678118654752440f));
        }
        xv[pr][i] = a;
      }
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int pr = 0; pr < 2; ++pr)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const float xf[8] = {xv[pr][2 * st][0], xv[pr][2 * st][1], xv[pr][2 * st][2], xv[pr][2 * st][3],
                           xv[pr][2 * st + 1][0], xv[pr][2 * st + 1][1], xv[pr][2 * st + 1][2], xv[pr][2 * st + 1][3]};
      const float wf[8] = {wv[pr][2 * st][0], wv[pr][2 * st][1], wv[pr][2 * st][2], wv[pr][2 * st][3],
                           wv[pr][2 * st + 1][0], wv[pr][2 * st + 1][1], wv[pr][2 * st + 1][2], wv[pr][2 * st + 1][3]};
      const HL a = split8(xf, p.single), b = split8(wf, p.single);
      if (!p.single) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, acc, 0, 0, 0);
    }
  // acc: lane = output column n0 + l31, register i = row m0 + crow(i, hf).  Sum the four waves' tiles through LDS.
#pragma unroll
  for (int i = 0; i < 16; ++i) sRed[wave][i][lane] = acc[i];
  __syncthreads();
  float* const plane_out = p.ws + (long long)kz * p.M * p.N;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = 4 * wave + q;
    const float v = (sRed[0][i][lane] + sRed[1][i][lane]) + (sRed[2][i][lane] + sRed[3][i][lane]);
    const int row = m0 + crow_t(i, hf);
    if (row < p.M) plane_out[(long long)row * p.N + n0 + l31] = v;
  }
}

}  // namespace

extern "C" int32_t tce_thin_linear_splits(int32_t M, int32_t N, int32_t K) {
  return (M > 0 && M <= 128 && N > 0 && N % 32 == 0 && K > 0 && K % 256 == 0 && K / 256 <= 64) ? K / 256 : -1;
}

extern "C" int tce_thin_partials_f32(const float* x, int64_t ldx, int32_t xsplits, const float* bias_x, int32_t act_x, const float* W,
                                     int64_t ldw, float* ws, int32_t M, int32_t N, int32_t K, tceStream stream) {
  TCE_CHECK_ARG(tce_thin_linear_splits(M, N, K) > 0, "tce_thin_partials_f32: unsupported shape M=%d N=%d K=%d (M <= 128, N %% 32, K %% 256)", M, N, K);
  TCE_CHECK_ARG(x && W && ws && tce_aligned16(x) && tce_aligned16(W) && tce_aligned16(ws) && (!bias_x || tce_aligned16(bias_x)),
                "tce_thin_partials_f32: null / misaligned pointer");
  TCE_CHECK_ARG(ldw >= K && ldw % 4 == 0 && xsplits >= 0 && xsplits <= 64 && (xsplits > 0 || (ldx >= K && ldx % 4 == 0)) &&
                    act_x >= 0 && act_x <= 2 && (xsplits > 0 || (!bias_x && act_x == 0)),
                "tce_thin_partials_f32: bad pitch / prologue arguments");
  TCE_CHECK_ARG(tce_get_gemm_mode() != 0, "tce_thin_partials_f32: split-fp16 arithmetic (GEMM modes 1 / 2); use tce_gemm_splitk_f32 in exact-fp32 mode");
  ThinArgs a;
  a.x = x; a.bias_x = bias_x; a.W = W; a.ws = ws; a.ldx = ldx; a.ldw = ldw; a.M = M; a.N = N; a.K = K;
  a.xsplits = xsplits; a.act_x = act_x; a.single = tce_gemm_single_pass();
  const dim3 grid(N / 32, tce_cdiv(M, 32), K / 256);
  if (xsplits > 0) hipLaunchKernelGGL(thin_partials_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(thin_partials_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  TCE_CHECK_LAUNCH("tce_thin_partials_f32");
  return TCE_OK;
}
