// Thin linear layers on the matrix cores: M <= a few dozen rows against a wide, deep weight matrix -- RoBERTa's dense
// layers at 32 tokens (tce_rvos.py:406-424: 768 x 2304 / 768 / 3072 and 3072 x 768), 28 MB of fp32 weights per encoder
// layer that are touched exactly once per clip.  Such a product is a weight STREAM: the only thing that matters is that
// every byte of W is requested early, by many workgroups, in one memory round trip.  The tiled GEMM's split-K form takes
// 13 us per projection here (a 64 x 64 tile half empty, a two-stage LDS pipeline of six short K slices: three dependent
// round trips); this kernel takes one:
//
//   grid (N/32, ceil(M/32), K/256); a workgroup = 32 output columns x 32 rows x 256 of K; each of its 4 waves loads ITS 64
//   of K of the W slab and of x in full 128-byte lines (8 lanes per row, sixteen 16-byte loads per lane, all in flight at
//   once), turns them into MFMA fragment order through a per-wave LDS tile, splits both to fp16 hi/lo on the fly (W is static
//   but streamed once: a pre-split copy would double its bytes), multiplies (x as A, W as B: output column on the lane),
//   the four waves' tiles are summed through LDS and the sums go to a partial plane ws[kz][M][N] in full 128-byte rows.
//   The partial planes (K/256 of them) meet in the reduction kernels of gemm.hip (bias, activation, residual, LayerNorm)
//   or in the NEXT consumer's load: PRO = the x operand is itself given as partial planes + bias (+ GELU) of the previous
//   layer (fc1 -> fc2 needs no reduction launch); tce_mha_small64_splits_f32 reads q, k, v the same way.
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

template <bool PRO, bool SINGLE>  // SINGLE: the arithmetic mode as a compile-time parameter (see ffn_fused_kernel, chain.hip)
__global__ void __launch_bounds__(256) thin_partials_kernel(const ThinArgs p) {
  // per wave: one staging tile for W and one for x (32 rows x 32 floats, 144-byte pitch: conflict-free for the row-of-8-lanes
  // writes and the row-per-lane fragment reads alike, see frag.h); the tiles are reused by the wave's two 32-wide K chunks
  __shared__ __attribute__((aligned(16))) float sT[4][2][32 * WT_PITCH];
  float (*const sRed)[16][64] = reinterpret_cast<float (*)[16][64]>(&sT[0][0][0]);  // [4][16][64] after the tiles are done (16 of 36 KB)
  static_assert(sizeof(float) * 4 * 16 * 64 <= sizeof(sT), "reduction buffer aliases the staging tiles");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hf = lane >> 5;
  const int cr = lane >> 3, cp = (lane & 7) * 4;  // coalesced side: row 8i + cr of the tile, 16-byte piece lane & 7
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32, kz = blockIdx.z;
  const int kb = kz * 256 + wave * 64;            // this wave's 64 of K: chunks kb .. kb+31 and kb+32 .. kb+63
  // every global load of the wave is issued before anything is consumed: ONE memory round trip (full 128-byte lines)
  f32x4 wv[2][4], xv[2][4];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      wv[c][i] = *reinterpret_cast<const f32x4*>(p.W + (long long)(n0 + 8 * i + cr) * p.ldw + kb + 32 * c + cp);
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = min(m0 + 8 * i + cr, p.M - 1);
      if (!PRO) {
        xv[c][i] = *reinterpret_cast<const f32x4*>(p.x + (long long)m * p.ldx + kb + 32 * c + cp);
      } else {
        const float* xr = p.x + (long long)m * p.K + kb + 32 * c + cp;
        const long long plane = (long long)p.M * p.K;
        f32x4 a = *reinterpret_cast<const f32x4*>(xr);
        for (int s = 1; s < p.xsplits; ++s) a += *reinterpret_cast<const f32x4*>(xr + s * plane);
        xv[c][i] = a;
      }
    }
  if (PRO) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f32x4 bb = {0.f, 0.f, 0.f, 0.f};
      if (p.bias_x) bb = *reinterpret_cast<const f32x4*>(p.bias_x + kb + 32 * c + cp);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 a = xv[c][i] + bb;
        if (p.act_x == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = fmaxf(a[e], 0.f);
        } else if (p.act_x == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = tce_gelu(a[e]);
        }
        xv[c][i] = a;
      }
    }
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float* const tw = &sT[wave][0][0];
  float* const tx = &sT[wave][1][0];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(tw + (8 * i + cr) * WT_PITCH + cp) = wv[c][i];
      *reinterpret_cast<f32x4*>(tx + (8 * i + cr) * WT_PITCH + cp) = xv[c][i];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int st = 0; st < 2; ++st) {  // fragment of k-step st: row l31, floats 16 st + 8 hf + 0..7 of the chunk
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(tx + l31 * WT_PITCH + 16 * st + 8 * hf);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(tx + l31 * WT_PITCH + 16 * st + 8 * hf + 4);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(tw + l31 * WT_PITCH + 16 * st + 8 * hf);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(tw + l31 * WT_PITCH + 16 * st + 8 * hf + 4);
      const float xf[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      const float wf[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
      const HL a = split8(xf, SINGLE), b = split8(wf, SINGLE);
      if (!SINGLE) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, acc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // acc: lane = output column n0 + l31, register i = row m0 + crow(i, hf).  Sum the four waves' tiles through LDS.
  __syncthreads();  // every wave is done with its staging tiles (the reduction buffer aliases them)
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
  if (a.single) {
    if (xsplits > 0) hipLaunchKernelGGL((thin_partials_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((thin_partials_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  } else {
    if (xsplits > 0) hipLaunchKernelGGL((thin_partials_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((thin_partials_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  TCE_CHECK_LAUNCH("tce_thin_partials_f32");
  return TCE_OK;
}
