// Row / group normalisations.  All HBM-bound: one read + one write of the activation, float4 accesses,
// wavefront-shuffle reductions (64 lanes), statistics in fp32 two-pass form (mean, then centred variance)
// to match PyTorch's numerics.
#include "common.h"
#include "../../include/tce_rvos.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// LayerNorm: one wavefront per row, rows of C floats (C % 4 == 0).  The row is read from HBM once
// (pass 1) and re-read from L1/L2 for the centred variance and the write.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ out,
                                                        long long M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * C);
  const f32x4* rr = r ? reinterpret_cast<const f32x4*>(r + row * C) : nullptr;
  const int n4 = C >> 2;
  float s = 0.f;
  for (int i = lane; i < n4; i += 64) {
    f32x4 v = xr[i];
    if (rr) v += rr[i];
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int i = lane; i < n4; i += 64) {
    f32x4 v = xr[i];
    if (rr) v += rr[i];
    const float a = v[0] - mean, b = v[1] - mean, c = v[2] - mean, d = v[3] - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
  const f32x4* b4 = reinterpret_cast<const f32x4*>(beta);
  f32x4* o4 = reinterpret_cast<f32x4*>(out + row * C);
  for (int i = lane; i < n4; i += 64) {
    f32x4 v = xr[i];
    if (rr) v += rr[i];
    const f32x4 g = g4[i], b = b4[i];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (v[j] - mean) * rstd * g[j] + b[j];
    o4[i] = o;
  }
}

// Rows of up to 256 floats (Swin stage 1 and every d_model = 256 tensor of the transformer): the row lives in ONE
// float4 per lane and is read once; LPR lanes per row (32 for C <= 128 -- the full-wave form above leaves 40 of 64
// lanes idle at C = 96 -- else 64), statistics by xor-shuffles inside those lanes.
template <int LPR>
__global__ void __launch_bounds__(256) layernorm_reg_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ out,
                                                            long long M, int C, float eps) {
  const int lane = threadIdx.x & 63, sub = lane & (LPR - 1);
  const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / LPR) + lane / LPR;
  const int n4 = C >> 2;
  const bool on = row < M && sub < n4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (on) {
    v = reinterpret_cast<const f32x4*>(x + row * C)[sub];
    if (r) v += reinterpret_cast<const f32x4*>(r + row * C)[sub];
  }
  float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  const float a = v[0] - mean, b = v[1] - mean, c = v[2] - mean, d = v[3] - mean;
  float q = on ? (a * a + b * b) + (c * c + d * d) : 0.f;
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)C + eps);
  if (on) {
    const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[sub], bb = reinterpret_cast<const f32x4*>(beta)[sub];
    f32x4 o;
    o[0] = a * rstd * g[0] + bb[0];
    o[1] = b * rstd * g[1] + bb[1];
    o[2] = c * rstd * g[2] + bb[2];
    o[3] = d * rstd * g[3] + bb[3];
    reinterpret_cast<f32x4*>(out + row * C)[sub] = o;
  }
}

// ---------------------------------------------------------------------------------------------------
// PatchMerging front half: virtual row = concat of 4 neighbours (zero outside), LayerNorm(4C).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) patch_merge_ln_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ out,
                                                             int T, int H, int W, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)T * H2 * W2) return;
  const int t = (int)(row / (H2 * W2));
  const int rem = (int)(row - (long long)t * H2 * W2);
  const int y2 = rem / W2, x2 = rem - y2 * W2;
  // quadrant order of the reference: (dy,dx) = (0,0),(1,0),(0,1),(1,1)
  const f32x4* src[4];
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const int yy = 2 * y2 + (qd & 1), xx = 2 * x2 + (qd >> 1);
    src[qd] = (yy < H && xx < W) ? reinterpret_cast<const f32x4*>(x + (((long long)t * H + yy) * W + xx) * C) : nullptr;
  }
  const int c4 = C >> 2, n4 = C;  // 4C floats = C float4
  float s = 0.f;
  for (int i = lane; i < n4; i += 64) {
    const int qd = i / c4, j = i - qd * c4;
    if (src[qd]) {
      const f32x4 v = src[qd][j];
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
  }
  const float mean = wave_sum(s) / (float)(4 * C);
  float q = 0.f;
  for (int i = lane; i < n4; i += 64) {
    const int qd = i / c4, j = i - qd * c4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src[qd]) v = src[qd][j];
    const float a = v[0] - mean, b = v[1] - mean, c = v[2] - mean, d = v[3] - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)(4 * C) + eps);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
  const f32x4* b4 = reinterpret_cast<const f32x4*>(beta);
  f32x4* o4 = reinterpret_cast<f32x4*>(out + row * 4 * C);
  for (int i = lane; i < n4; i += 64) {
    const int qd = i / c4, j = i - qd * c4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src[qd]) v = src[qd][j];
    const f32x4 g = g4[i], b = b4[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (v[k] - mean) * rstd * g[k] + b[k];
    o4[i] = o;
  }
}

// ---------------------------------------------------------------------------------------------------
// PatchEmbed: 4x4 stride-4 conv of an NCHW frame (3 channels -> K = 48) + LayerNorm(C).
// One wavefront per output token: the 48 input taps are wave-uniform (LDS broadcast), lane c computes
// channels c, c+64, ...  Weights w [C,48] are read once per workgroup into LDS.
// ---------------------------------------------------------------------------------------------------
template <int MAXC_PER_LANE>
__global__ void __launch_bounds__(256) patch_embed_kernel(const float* __restrict__ frames,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ out,
                                                          int T, int H, int W, int C, float eps, int tokens_per_block) {
  extern __shared__ float sm[];
  float* ws = sm;                 // [48][C]  (k-major so that lanes read consecutive channels)
  float* taps = sm + 48 * C;      // [4 waves][48]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 48 * C; i += 256) {
    const int c = i / 48, k = i - c * 48;
    ws[k * C + c] = w[i];
  }
  __syncthreads();
  const int Hp = (H + 3) >> 2, Wp = (W + 3) >> 2;
  const long long ntok = (long long)T * Hp * Wp;
  for (int it = 0; it < tokens_per_block; it += 4) {
    const long long tok = (long long)blockIdx.x * tokens_per_block + it + wave;
    if (tok < ntok) {
      const int t = (int)(tok / (Hp * Wp));
      const int rem = (int)(tok - (long long)t * Hp * Wp);
      const int py = rem / Wp, px = rem - py * Wp;
      if (lane < 48) {
        const int c = lane >> 4, ky = (lane >> 2) & 3, kx = lane & 3;
        const int yy = py * 4 + ky, xx = px * 4 + kx;
        taps[wave * 48 + lane] = (yy < H && xx < W) ? frames[(((long long)t * 3 + c) * H + yy) * W + xx] : 0.f;
      }
    }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    if (tok < ntok) {
      float acc[MAXC_PER_LANE];
#pragma unroll
      for (int j = 0; j < MAXC_PER_LANE; ++j) {
        const int c = lane + 64 * j;
        acc[j] = (c < C) ? b[c] : 0.f;
      }
      for (int k = 0; k < 48; ++k) {
        const float xv = taps[wave * 48 + k];
#pragma unroll
        for (int j = 0; j < MAXC_PER_LANE; ++j) {
          const int c = lane + 64 * j;
          if (c < C) acc[j] = fmaf(xv, ws[k * C + c], acc[j]);
        }
      }
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < MAXC_PER_LANE; ++j)
        if (lane + 64 * j < C) s += acc[j];
      const float mean = wave_sum(s) / (float)C;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < MAXC_PER_LANE; ++j)
        if (lane + 64 * j < C) {
          const float d = acc[j] - mean;
          q += d * d;
        }
      const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
      for (int j = 0; j < MAXC_PER_LANE; ++j) {
        const int c = lane + 64 * j;
        if (c < C) out[tok * C + c] = (acc[j] - mean) * rstd * gamma[c] + beta[c];
      }
    }
    __syncthreads();
  }
}

// Lane-per-token form (C % 8 == 0, C <= 152; TOK tokens x (C+4) floats of LDS): a lane gathers its
// token's 48 taps with 16-byte loads (lanes = consecutive patches of a row -> coalesced), the weights are wave-uniform
// (scalar loads), the raw conv outputs go to an LDS row per token, LayerNorm statistics are a serial pass over the
// lane's own row (no cross-lane traffic), and the normalised tile -- TOK tokens x C floats, contiguous in the
// token-major output -- is written cooperatively as a float4 stream.
template <int CO_CHUNK, int TOK>
__global__ void __launch_bounds__(TOK) patch_embed_lane_kernel(const float* __restrict__ frames,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ out,
                                                               const int H, const int W, const int C, const float eps,
                                                               const long long ntok, const int Hp, const int Wp) {
  extern __shared__ float sm[];
  const int pitch = C + 4;
  float* raw = sm;                      // [TOK][pitch]
  float* smean = sm + TOK * pitch;      // [TOK]
  float* srstd = smean + TOK;           // [TOK]
  const int tid = threadIdx.x;
  const long long tok0 = (long long)blockIdx.x * TOK;
  const long long tok = tok0 + tid;
  float tap[48];
  if (tok < ntok) {
    const int t = (int)(tok / (Hp * Wp));
    const int rem = (int)(tok - (long long)t * Hp * Wp);
    const int py = rem / Wp, px = rem - py * Wp;
    const bool vec = ((W & 3) == 0) && (px * 4 + 3 < W);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 4; ++ky) {
        const int yy = py * 4 + ky;
        const float* row = frames + (((long long)t * 3 + c) * H + min(yy, H - 1)) * W;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (yy < H) {
          if (vec) {
            v = *reinterpret_cast<const f32x4*>(row + px * 4);
          } else {
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
              if (px * 4 + kx < W) v[kx] = row[px * 4 + kx];
          }
        }
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) tap[c * 16 + ky * 4 + kx] = v[kx];
      }
  } else {
#pragma unroll
    for (int k = 0; k < 48; ++k) tap[k] = 0.f;
  }
  float* myrow = raw + tid * pitch;
  float s = 0.f;
  for (int co = 0; co < C; co += CO_CHUNK) {  // weights are wave-uniform: the compiler keeps them in SGPRs
    float acc[CO_CHUNK];
#pragma unroll
    for (int j = 0; j < CO_CHUNK; ++j) acc[j] = b[co + j];
#pragma unroll
    for (int j = 0; j < CO_CHUNK; ++j) {
      const float* wr = w + (long long)(co + j) * 48;
#pragma unroll
      for (int k = 0; k < 48; ++k) acc[j] = fmaf(tap[k], wr[k], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < CO_CHUNK; j += 4) {
      f32x4 v = {acc[j], acc[j + 1], acc[j + 2], acc[j + 3]};
      *reinterpret_cast<f32x4*>(myrow + co + j) = v;
      s += (acc[j] + acc[j + 1]) + (acc[j + 2] + acc[j + 3]);
    }
  }
  const float mean = s / (float)C;
  float q = 0.f;
  for (int c = 0; c < C; c += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(myrow + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[e] - mean;
      q = fmaf(d, d, q);
    }
  }
  smean[tid] = mean;
  srstd[tid] = rsqrtf(q / (float)C + eps);
  __syncthreads();
  const int c4n = C >> 2;
  const long long nvalid = min((long long)TOK, ntok - tok0);
  f32x4* o4 = reinterpret_cast<f32x4*>(out + tok0 * C);
  for (long long i = tid; i < nvalid * c4n; i += TOK) {
    const int r = (int)(i / c4n), c4 = (int)(i - (long long)r * c4n);
    const f32x4 v = *reinterpret_cast<const f32x4*>(raw + r * pitch + c4 * 4);
    const f32x4 g4 = reinterpret_cast<const f32x4*>(gamma)[c4], b4 = reinterpret_cast<const f32x4*>(beta)[c4];
    const float m = smean[r], rs = srstd[r];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[e] - m) * rs * g4[e] + b4[e];
    o4[i] = o;
  }
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm on channels-last [T, HW, C]: each (frame, group) is split over `nsplit` workgroups that
// compute (count, mean, M2) of their row range (two passes over the chunk, second one from L2); the apply
// kernel merges the partials with Chan's formula and normalises (+ReLU).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) groupnorm_stats_kernel(const float* __restrict__ x, float* __restrict__ ws,
                                                              int HW, int C, int G, int nsplit) {
  __shared__ float red[4];
  const int tg = blockIdx.x;  // t*G + g
  const int sp = blockIdx.y;
  const int t = tg / G, g = tg - t * G;
  const int cg = C / G;
  const int rows_per = (HW + nsplit - 1) / nsplit;
  const int r0 = sp * rows_per, r1 = min(HW, r0 + rows_per);
  const float* base = x + (long long)t * HW * C + g * cg;
  const int n = max(0, r1 - r0) * cg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f;
  for (int i = tid; i < n; i += 256) {
    const int rr = i / cg, c = i - rr * cg;
    s += base[(long long)(r0 + rr) * C + c];
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mean = n > 0 ? ((red[0] + red[1]) + (red[2] + red[3])) / (float)n : 0.f;
  __syncthreads();
  float q = 0.f;
  for (int i = tid; i < n; i += 256) {
    const int rr = i / cg, c = i - rr * cg;
    const float d = base[(long long)(r0 + rr) * C + c] - mean;
    q += d * d;
  }
  q = wave_sum(q);
  if (lane == 0) red[wave] = q;
  __syncthreads();
  if (tid == 0) {
    float* o = ws + ((long long)tg * nsplit + sp) * 3;
    o[0] = (float)n;
    o[1] = mean;
    o[2] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// Single-pass form for C = 256 (every GroupNorm of the path: tce_rvos.py:81,86 with 32 groups, segmentation.py:43 with 8):
// a wave reads one full 1 KiB row per instruction (64 lanes x 16 B, lane = 4 consecutive channels, all of one group),
// a workgroup keeps its 4 x R rows in registers, so the chunk's mean and its M2 about that mean (the same two-pass
// arithmetic as above) cost ONE trip to HBM; 4 x R float4 loads are in flight per lane.
template <int R>
__global__ void __launch_bounds__(256) groupnorm_stats_rows_kernel(const float* __restrict__ x, float* __restrict__ ws,
                                                                   const int HW, const int G, const int rows_per,
                                                                   const int nsplit) {
  __shared__ float red[4][64];
  const int t = blockIdx.y, sp = blockIdx.x;
  const int r0 = sp * rows_per, r1 = min(HW, r0 + rows_per);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lpg = 64 / G;  // lanes per group: 2 (32 groups) or 8 (8 groups)
  const float* base = x + (long long)t * HW * 256 + lane * 4;
  f32x4 v[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int row = r0 + wave + 4 * i;
    v[i] = *reinterpret_cast<const f32x4*>(base + (long long)min(row, HW - 1) * 256);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < R; ++i)
    if (r0 + wave + 4 * i < r1) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  for (int o = 1; o < lpg; o <<= 1) s += __shfl_xor(s, o, 64);
  red[wave][lane] = s;
  __syncthreads();
  const float n = (float)((r1 - r0) * (256 / G));
  const float mean = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) / n;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < R; ++i)
    if (r0 + wave + 4 * i < r1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        q = fmaf(d, d, q);
      }
    }
  for (int o = 1; o < lpg; o <<= 1) q += __shfl_xor(q, o, 64);
  red[wave][lane] = q;
  __syncthreads();
  if (wave == 0 && (lane % lpg) == 0) {
    const int g = lane / lpg;
    float* o = ws + ((long long)(t * G + g) * nsplit + sp) * 3;
    o[0] = n;
    o[1] = mean;
    o[2] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  }
}

// merges partial (count, mean, M2) triples of a (frame, group) with Chan's formula (lanes take partials lane, lane + 16, ...:
// independent L2 round trips -- a serial walk over ~100 partials would cost more than the statistics pass -- then a
// butterfly merges the lanes)
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, const float nb, const float mb, const float m2b) {
  const float nt = n + nb;
  if (nt > 0.f) {
    const float d = mb - mean, f = nb / nt;
    mean += d * f;
    m2 += m2b + d * d * (n * f);
  }
  n = nt;
}
// scale[c] = rstd * gamma[c], shift[c] = beta[c] - mean * scale[c] of frame t into LDS; the merge of the per-chunk partials is done
// here (see groupnorm_apply_kernel).  Ends with a workgroup barrier.
__device__ __forceinline__ void groupnorm_scale_shift(const float* __restrict__ ws, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* scale, float* shift, const int t,
                                                      const int C, const int G, const int nsplit, const float eps) {
  __shared__ float sMean[64], sRstd[64];
  const int cg = C / G;
  {
    const int l16 = threadIdx.x & 15;
    for (int g = threadIdx.x >> 4; g < G; g += 16) {
      const float* p = ws + (long long)(t * G + g) * nsplit * 3;
      float n = 0.f, mean = 0.f, m2 = 0.f;
      for (int i0 = l16; i0 < nsplit; i0 += 64) {  // four independent partials per step
        float pn[4], pm[4], p2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + 16 * u;
          const bool ok = i < nsplit;
          pn[u] = ok ? p[3 * i] : 0.f;
          pm[u] = ok ? p[3 * i + 1] : 0.f;
          p2[u] = ok ? p[3 * i + 2] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) chan_merge(n, mean, m2, pn[u], pm[u], p2[u]);
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), m2b = __shfl_xor(m2, o, 64);
        chan_merge(n, mean, m2, nb, mb, m2b);
      }
      if (l16 == 0) {
        sMean[g] = mean;
        sRstd[g] = rsqrtf(m2 / n + eps);
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cg;
    const float mean = sMean[g], rstd = sRstd[g];
    const float sc = rstd * gamma[c];
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
  }
  __syncthreads();
}

// Applies the normalisation.  The merge of the per-chunk partials ("finalize") is done HERE, by every workgroup for its
// frame's G groups (16 lanes per group: a lane's partials are independent loads, then a 4-step butterfly of Chan's
// formula): G * nsplit * 12 bytes from L2 per workgroup instead of a separate 5 us launch on the pixel decoder's critical
// path (12 per clip).
__global__ void __launch_bounds__(256) groupnorm_apply_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ ws,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ out,
                                                              int HW, int C, int G, int relu, int rows_per_block,
                                                              int nsplit, float eps) {
  extern __shared__ float sm[];  // scale[C], shift[C]
  float* scale = sm;
  float* shift = sm + C;
  const int t = blockIdx.y;
  groupnorm_scale_shift(ws, gamma, beta, scale, shift, t, C, G, nsplit, eps);
  const int c4 = C >> 2;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min((long long)HW, r0 + rows_per_block);
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x + (long long)t * HW * C);
  f32x4* o4 = reinterpret_cast<f32x4*>(out + (long long)t * HW * C);
  const f32x4* sc4 = reinterpret_cast<const f32x4*>(scale);
  const f32x4* sh4 = reinterpret_cast<const f32x4*>(shift);
  for (long long i = r0 * c4 + threadIdx.x; i < r1 * c4; i += 256) {
    const int c = (int)(i % c4);
    const f32x4 v = x4[i], a = sc4[c], b = sh4[c];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = v[j] * a[j] + b[j];
      if (relu) o[j] = fmaxf(o[j], 0.f);
    }
    o4[i] = o;
  }
}

// The same, written through a nearest-neighbour up-sampling onto a finer map and added to it (round 5):
//     out[t, yo, xo, :] = add[t, yo, xo, :] + act(GN(x)[t, yi, xi, :]),   yi = min(floor(yo * h / ho), h - 1) (F.interpolate "nearest")
// -- the pixel decoder's top-down merge `cur_fpn + interpolate(y)` with y = ReLU(GN(conv)) (segmentation.py:199-203): the
// normalised coarse map is never written, and the merge is not a launch of its own.  Same arithmetic as the two launches
// (normalise, ReLU, then + add).  out may alias add.
__global__ void __launch_bounds__(256) groupnorm_apply_up_kernel(const float* __restrict__ x, const float* __restrict__ ws,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* add, float* out, int h, int w, int ho, int wo,
                                                                 int C, int G, int relu, int rows_per_block, int nsplit, float eps) {
  extern __shared__ float sm[];  // scale[C], shift[C]
  float* scale = sm;
  float* shift = sm + C;
  const int t = blockIdx.y;
  groupnorm_scale_shift(ws, gamma, beta, scale, shift, t, C, G, nsplit, eps);
  const int c4 = C >> 2;
  const long long HWo = (long long)ho * wo;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(HWo, r0 + rows_per_block);
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x + (long long)t * h * w * C);
  const f32x4* a4 = reinterpret_cast<const f32x4*>(add + (long long)t * HWo * C);
  f32x4* o4 = reinterpret_cast<f32x4*>(out + (long long)t * HWo * C);
  const f32x4* sc4 = reinterpret_cast<const f32x4*>(scale);
  const f32x4* sh4 = reinterpret_cast<const f32x4*>(shift);
  const float sy = (float)h / (float)ho, sx = (float)w / (float)wo;
  for (long long i = r0 * c4 + threadIdx.x; i < r1 * c4; i += 256) {
    const int c = (int)(i % c4);
    const int ro = (int)(i / c4);
    const int yo = ro / wo, xo = ro - yo * wo;
    const int yi = min((int)floorf((float)yo * sy), h - 1), xi = min((int)floorf((float)xo * sx), w - 1);
    const f32x4 v = x4[((long long)yi * w + xi) * c4 + c], a = sc4[c], b = sh4[c];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = v[j] * a[j] + b[j];
      if (relu) o[j] = fmaxf(o[j], 0.f);
    }
    o += a4[i];
    o4[i] = o;
  }
}


}  // namespace

extern "C" int tce_layernorm_f32(const float* x, const float* r, const float* gamma, const float* beta, float* out,
                                 int64_t M, int32_t C, float eps, tceStream stream) {
  TCE_CHECK_ARG(x && gamma && beta && out, "tce_layernorm_f32: null pointer");
  TCE_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "tce_layernorm_f32: need M>0, C%%4==0 (M=%lld C=%d)", (long long)M, C);
  TCE_CHECK_ARG(tce_aligned16(x) && tce_aligned16(out) && tce_aligned16(gamma) && tce_aligned16(beta) &&
                    (!r || tce_aligned16(r)),
                "tce_layernorm_f32: pointers must be 16-byte aligned");
  if (C <= 128)
    hipLaunchKernelGGL(layernorm_reg_kernel<32>, dim3(tce_cdiv(M, 8)), dim3(256), 0, (hipStream_t)stream, x, r, gamma, beta,
                       out, (long long)M, C, eps);
  else if (C <= 256)
    hipLaunchKernelGGL(layernorm_reg_kernel<64>, dim3(tce_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, r, gamma, beta,
                       out, (long long)M, C, eps);
  else
    hipLaunchKernelGGL(layernorm_kernel, dim3(tce_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, r, gamma, beta, out,
                     (long long)M, C, eps);
  TCE_CHECK_LAUNCH("tce_layernorm_f32");
  return TCE_OK;
}

extern "C" int tce_patch_merge_ln_f32(const float* x, const float* gamma, const float* beta, float* out, int32_t T,
                                      int32_t H, int32_t W, int32_t C, float eps, tceStream stream) {
  TCE_CHECK_ARG(x && gamma && beta && out, "tce_patch_merge_ln_f32: null pointer");
  TCE_CHECK_ARG(T > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "tce_patch_merge_ln_f32: bad shape");
  TCE_CHECK_ARG(tce_aligned16(x) && tce_aligned16(out) && tce_aligned16(gamma) && tce_aligned16(beta),
                "tce_patch_merge_ln_f32: pointers must be 16-byte aligned");
  const long long rows = (long long)T * ((H + 1) / 2) * ((W + 1) / 2);
  hipLaunchKernelGGL(patch_merge_ln_kernel, dim3(tce_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                     out, T, H, W, C, eps);
  TCE_CHECK_LAUNCH("tce_patch_merge_ln_f32");
  return TCE_OK;
}

extern "C" int tce_patch_embed_f32(const float* frames, const float* w, const float* b, const float* gamma,
                                   const float* beta, float* out, int32_t T, int32_t H, int32_t W, int32_t C, float eps,
                                   tceStream stream) {
  TCE_CHECK_ARG(frames && w && b && gamma && beta && out, "tce_patch_embed_f32: null pointer");
  TCE_CHECK_ARG(T > 0 && H > 0 && W > 0 && C > 0 && C <= 256, "tce_patch_embed_f32: need 0 < C <= 256 (C=%d)", C);
  const int Hp = (H + 3) / 4, Wp = (W + 3) / 4;
  const long long ntok = (long long)T * Hp * Wp;
  // split-fp16 / fp16 GEMM modes: the taps go through the matrix cores (csrc/chain.hip); exact-fp32 mode keeps the
  // fp32 vector kernels below
  if (tce_get_gemm_mode() != 0 &&
      tce_patch_embed_mfma(frames, w, b, gamma, beta, out, H, W, C, eps, ntok, Hp, Wp, (hipStream_t)stream)) {
    TCE_CHECK_LAUNCH("tce_patch_embed_f32");
    return TCE_OK;
  }
  // (C <= 120: the lane kernel's dynamic LDS stays below the 64 KB a launch may request without raising the function's
  // limit; wider embeddings take the generic kernel below.)
  if (C % 8 == 0 && C <= 120 && tce_aligned16(frames) && tce_aligned16(out) && tce_aligned16(gamma) && tce_aligned16(beta)) {
    constexpr int TOK = 128;  // tokens (= threads) per workgroup: 52 KiB of LDS at C = 96 -> three workgroups per CU
    const size_t lds = (size_t)(TOK * (C + 4) + 2 * TOK) * sizeof(float);
    hipLaunchKernelGGL((patch_embed_lane_kernel<8, TOK>), dim3(tce_cdiv(ntok, TOK)), dim3(TOK), lds, (hipStream_t)stream, frames, w, b,
                       gamma, beta, out, H, W, C, eps, ntok, Hp, Wp);
    TCE_CHECK_LAUNCH("tce_patch_embed_f32");
    return TCE_OK;
  }
  const int tpb = 64;
  const size_t smem = (size_t)(48 * C + 4 * 48) * sizeof(float);
  hipLaunchKernelGGL((patch_embed_kernel<4>), dim3(tce_cdiv(ntok, tpb)), dim3(256), smem, (hipStream_t)stream, frames, w,
                     b, gamma, beta, out, T, H, W, C, eps, tpb);
  TCE_CHECK_LAUNCH("tce_patch_embed_f32");
  return TCE_OK;
}

static int groupnorm_rows_per(int HW) { return HW >= 4096 ? 128 : (HW >= 512 ? 32 : 8); }
extern "C" int tce_groupnorm_nsplit(int32_t HW) { return tce_cdiv(HW, groupnorm_rows_per(HW)); }

static void groupnorm_stats_launch(const float* x, float* ws, int T, int HW, int C, int G, hipStream_t s);

extern "C" int tce_groupnorm_up_add_f32(const float* x, const float* gamma, const float* beta, const float* add, float* out,
                                        float* ws, int32_t T, int32_t h, int32_t w, int32_t ho, int32_t wo, int32_t C, int32_t G,
                                        float eps, int32_t relu, tceStream stream) {
  TCE_CHECK_ARG(x && gamma && beta && add && out && ws, "tce_groupnorm_up_add_f32: null pointer");
  TCE_CHECK_ARG(T > 0 && h > 0 && w > 0 && ho > 0 && wo > 0 && C > 0 && G > 0 && G <= 64 && C % G == 0 && C % 4 == 0,
                "tce_groupnorm_up_add_f32: bad shape (G <= 64)");
  TCE_CHECK_ARG(tce_aligned16(x) && tce_aligned16(out) && tce_aligned16(add), "tce_groupnorm_up_add_f32: x/add/out must be 16-byte aligned");
  TCE_CHECK_ARG(x != out, "tce_groupnorm_up_add_f32: out may alias add, not x");
  const int HW = h * w, nsplit = tce_cdiv(HW, groupnorm_rows_per(HW));
  hipStream_t s = (hipStream_t)stream;
  groupnorm_stats_launch(x, ws, T, HW, C, G, s);
  const int rows_per_block = 64;
  hipLaunchKernelGGL(groupnorm_apply_up_kernel, dim3(tce_cdiv(ho * wo, rows_per_block), T), dim3(256), (size_t)2 * C * sizeof(float), s,
                     x, ws, gamma, beta, add, out, h, w, ho, wo, C, G, relu, rows_per_block, nsplit, eps);
  TCE_CHECK_LAUNCH("tce_groupnorm_up_add_f32");
  return TCE_OK;
}

static void groupnorm_stats_launch(const float* x, float* ws, int T, int HW, int C, int G, hipStream_t s) {
  const int rows_per = groupnorm_rows_per(HW), nsplit = tce_cdiv(HW, rows_per);
  if (C == 256 && (G == 8 || G == 16 || G == 32 || G == 64)) {
    const dim3 grid(nsplit, T);
    if (rows_per == 128) hipLaunchKernelGGL(groupnorm_stats_rows_kernel<32>, grid, dim3(256), 0, s, x, ws, HW, G, rows_per, nsplit);
    else if (rows_per == 32) hipLaunchKernelGGL(groupnorm_stats_rows_kernel<8>, grid, dim3(256), 0, s, x, ws, HW, G, rows_per, nsplit);
    else hipLaunchKernelGGL(groupnorm_stats_rows_kernel<2>, grid, dim3(256), 0, s, x, ws, HW, G, rows_per, nsplit);
  } else {
    hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(T * G, nsplit), dim3(256), 0, s, x, ws, HW, C, G, nsplit);
  }
}

extern "C" int tce_groupnorm_f32(const float* x, const float* gamma, const float* beta, float* out, float* ws, int32_t T,
                                 int32_t HW, int32_t C, int32_t G, float eps, int32_t relu, tceStream stream) {
  TCE_CHECK_ARG(x && gamma && beta && out && ws, "tce_groupnorm_f32: null pointer");
  TCE_CHECK_ARG(T > 0 && HW > 0 && C > 0 && G > 0 && G <= 64 && C % G == 0 && C % 4 == 0, "tce_groupnorm_f32: bad shape (G <= 64)");
  TCE_CHECK_ARG(tce_aligned16(x) && tce_aligned16(out), "tce_groupnorm_f32: x/out must be 16-byte aligned");
  const int rows_per = groupnorm_rows_per(HW), nsplit = tce_cdiv(HW, rows_per);
  hipStream_t s = (hipStream_t)stream;
  if (C == 256 && (G == 8 || G == 16 || G == 32 || G == 64)) {
    const dim3 grid(nsplit, T);
    if (rows_per == 128) hipLaunchKernelGGL(groupnorm_stats_rows_kernel<32>, grid, dim3(256), 0, s, x, ws, HW, G, rows_per, nsplit);
    else if (rows_per == 32) hipLaunchKernelGGL(groupnorm_stats_rows_kernel<8>, grid, dim3(256), 0, s, x, ws, HW, G, rows_per, nsplit);
    else hipLaunchKernelGGL(groupnorm_stats_rows_kernel<2>, grid, dim3(256), 0, s, x, ws, HW, G, rows_per, nsplit);
  } else {
    hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(T * G, nsplit), dim3(256), 0, s, x, ws, HW, C, G, nsplit);
  }
  const int rows_per_block = 64;
  // the apply kernel merges the partials itself (no separate finalize launch)
  hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(tce_cdiv(HW, rows_per_block), T), dim3(256),
                     (size_t)2 * C * sizeof(float), s, x, ws, gamma, beta, out, HW, C, G, relu, rows_per_block, nsplit, eps);
  TCE_CHECK_LAUNCH("tce_groupnorm_f32");
  return TCE_OK;
}
