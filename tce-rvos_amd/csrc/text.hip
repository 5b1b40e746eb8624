// RoBERTa text encoder pieces that the GEMM / LayerNorm kernels do not already cover
// (reference call site: models/tce_rvos.py:406-424; arithmetic: HuggingFace RobertaModel).
#include "common.h"
#include "../../include/tce_rvos.h"

namespace {

// embeddings: word[ids] + position[pos_ids] + token_type[0] -> LayerNorm.  One wavefront per token.
__global__ void __launch_bounds__(256) embed_ln_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pos_ids,
                                                       const float* __restrict__ word, const float* __restrict__ pos,
                                                       const float* __restrict__ type0, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ out, int L,
                                                       int C, float eps, int pad_id, int seq_len) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= L) return;
  long long pid;
  if (pos_ids) {
    pid = pos_ids[tok];
  } else {
    // HF create_position_ids_from_input_ids: pad + (number of non-pad tokens up to and including this one), pad -> pad;
    // seq_len: the L tokens are L / seq_len captions of equal length, the count restarts at each caption
    int cnt = 0;
    for (int i = (tok / seq_len) * seq_len + lane; i <= tok; i += 64) cnt += ids[i] != pad_id;
    cnt = (int)wave_sum((float)cnt);
    pid = ids[tok] != pad_id ? pad_id + cnt : pad_id;
  }
  const f32x4* w4 = reinterpret_cast<const f32x4*>(word + ids[tok] * C);
  const f32x4* p4 = reinterpret_cast<const f32x4*>(pos + pid * C);
  const f32x4* t4 = reinterpret_cast<const f32x4*>(type0);
  const int n4 = C >> 2;
  float s = 0.f;
  for (int i = lane; i < n4; i += 64) {
    const f32x4 v = w4[i] + t4[i] + p4[i];
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int i = lane; i < n4; i += 64) {
    const f32x4 v = w4[i] + t4[i] + p4[i];
    const float a = v[0] - mean, b = v[1] - mean, c = v[2] - mean, d = v[3] - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
  const f32x4* b4 = reinterpret_cast<const f32x4*>(beta);
  f32x4* o4 = reinterpret_cast<f32x4*>(out + (long long)tok * C);
  for (int i = lane; i < n4; i += 64) {
    const f32x4 v = w4[i] + t4[i] + p4[i], g = g4[i], b = b4[i];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (v[j] - mean) * rstd * g[j] + b[j];
    o4[i] = o;
  }
}

// Short-sequence self-attention with head_dim 64 (12 heads x 64 for RoBERTa-base).  One workgroup per head; K and V
// of the whole sentence (L <= 128 tokens) live in LDS; lane = query row.
// splits > 1 / bias: qkv is given as `splits` partial planes [L, 3E] (split-K partial sums, plane stride L*3E) + a bias [3E]:
// the reduction of the projection rides in this kernel's loads (no reduce launch between projection and attention).
template <int HDIM, int LMAX>
__global__ void __launch_bounds__(256) mha_small_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L,
                                                        int nheads, float scale, int splits, const float* __restrict__ bias) {
  // gridDim.z sequences of L tokens stacked as rows: sequence z owns rows z*L .. z*L + L - 1 of every plane and of `out`
  __shared__ __attribute__((aligned(16))) float sK[LMAX * HDIM];
  __shared__ __attribute__((aligned(16))) float sV[LMAX * HDIM];
  const int h = blockIdx.x;
  const int E = nheads * HDIM;
  const int tid = threadIdx.x;
  const long long plane = (long long)L * gridDim.z * 3 * E;
  qkv += (long long)blockIdx.z * L * 3 * E;
  out += (long long)blockIdx.z * L * E;
  auto ld4 = [&](const float* p, int col) {  // sum of the partial planes (+ bias) at p
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4*>(p + s * plane);
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + col);
    return v;
  };
  // 256 threads stage K / V (with partial planes that is three times the loads); the upper 128 leave after the barrier
  for (int i = tid; i < L * (HDIM / 4); i += 256) {
    const int j = i / (HDIM / 4), d4 = i % (HDIM / 4);
    const float* p = qkv + (long long)j * 3 * E + h * HDIM + d4 * 4;
    *reinterpret_cast<f32x4*>(&sK[j * HDIM + d4 * 4]) = ld4(p + E, E + h * HDIM + d4 * 4);
    *reinterpret_cast<f32x4*>(&sV[j * HDIM + d4 * 4]) = ld4(p + 2 * E, 2 * E + h * HDIM + d4 * 4);
  }
  __syncthreads();
  if (tid >= 128) return;
  // four lanes per query (a quarter of the head dim each): the score is a 16-term partial dot + two shuffles, the output
  // row is split the same way -- a thread per query walked the keys with 128 dependent FMAs each (19 us for 32 tokens)
  constexpr int QD = HDIM / 4;
  const int i = blockIdx.y * 32 + (tid >> 2), part = tid & 3;
  const int ic = min(i, L - 1);
  float q[QD];
  {
    const float* p = qkv + (long long)ic * 3 * E + h * HDIM + part * QD;
#pragma unroll
    for (int d4 = 0; d4 < QD / 4; ++d4) {
      const f32x4 v = ld4(p + d4 * 4, h * HDIM + part * QD + d4 * 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) q[d4 * 4 + c] = v[c] * scale;
    }
  }
  float m = -3.0e38f, l = 0.f;
  float o[QD];
#pragma unroll
  for (int d = 0; d < QD; ++d) o[d] = 0.f;
  for (int j = 0; j < L; ++j) {  // online softmax, one key at a time (L is tiny)
    const f32x4* kp = reinterpret_cast<const f32x4*>(&sK[j * HDIM + part * QD]);
    float a = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < QD / 4; ++d4) {
      const f32x4 kv = kp[d4];
      a = fmaf(q[d4 * 4 + 0], kv[0], a);
      a = fmaf(q[d4 * 4 + 1], kv[1], a);
      a = fmaf(q[d4 * 4 + 2], kv[2], a);
      a = fmaf(q[d4 * 4 + 3], kv[3], a);
    }
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    const float mnew = fmaxf(m, a);
    const float corr = __expf(m - mnew), pj = __expf(a - mnew);
    l = l * corr + pj;
    const f32x4* vp = reinterpret_cast<const f32x4*>(&sV[j * HDIM + part * QD]);
#pragma unroll
    for (int d4 = 0; d4 < QD / 4; ++d4) {
      const f32x4 vv = vp[d4];
#pragma unroll
      for (int c = 0; c < 4; ++c) o[d4 * 4 + c] = fmaf(pj, vv[c], o[d4 * 4 + c] * corr);
    }
    m = mnew;
  }
  if (i >= L) return;
  const float inv = 1.f / l;
  float* po = out + (long long)i * E + h * HDIM + part * QD;
#pragma unroll
  for (int d4 = 0; d4 < QD / 4; ++d4) {
    f32x4 v = {o[d4 * 4] * inv, o[d4 * 4 + 1] * inv, o[d4 * 4 + 2] * inv, o[d4 * 4 + 3] * inv};
    *reinterpret_cast<f32x4*>(po + d4 * 4) = v;
  }
}

__global__ void __launch_bounds__(256) tanh_kernel(const float* __restrict__ x, float* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = tanhf(x[i]);
}

}  // namespace

extern "C" int tce_embed_ln_f32(const int64_t* ids, const int64_t* pos_ids, const float* word, const float* pos,
                                const float* type0, const float* gamma, const float* beta, float* out, int32_t L,
                                int32_t C, float eps, int32_t pad_id, tceStream stream) {
  TCE_CHECK_ARG(ids && word && pos && type0 && gamma && beta && out && L > 0 && C > 0 && C % 4 == 0,
                "tce_embed_ln_f32: bad arguments");
  hipLaunchKernelGGL(embed_ln_kernel, dim3(tce_cdiv(L, 4)), dim3(256), 0, (hipStream_t)stream, ids, pos_ids, word, pos,
                     type0, gamma, beta, out, L, C, eps, pad_id, L);
  TCE_CHECK_LAUNCH("tce_embed_ln_f32");
  return TCE_OK;
}

extern "C" int tce_embed_ln_seqs_f32(const int64_t* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                                     const float* beta, float* out, int32_t nseq, int32_t seq_len, int32_t C, float eps, int32_t pad_id,
                                     tceStream stream) {
  TCE_CHECK_ARG(ids && word && pos && type0 && gamma && beta && out && nseq > 0 && seq_len > 0 && C > 0 && C % 4 == 0,
                "tce_embed_ln_seqs_f32: bad arguments");
  const int L = nseq * seq_len;
  hipLaunchKernelGGL(embed_ln_kernel, dim3(tce_cdiv(L, 4)), dim3(256), 0, (hipStream_t)stream, ids, (const int64_t*)nullptr, word, pos,
                     type0, gamma, beta, out, L, C, eps, pad_id, seq_len);
  TCE_CHECK_LAUNCH("tce_embed_ln_seqs_f32");
  return TCE_OK;
}

extern "C" int tce_mha_small64_f32(const float* qkv, float* out, int32_t L, int32_t nheads, float scale,
                                   tceStream stream) {
  TCE_CHECK_ARG(qkv && out && nheads > 0, "tce_mha_small64_f32: bad arguments");
  TCE_CHECK_ARG(L > 0 && L <= 128, "tce_mha_small64_f32: sequence length %d outside 1..128", L);
  TCE_CHECK_ARG(tce_aligned16(qkv) && tce_aligned16(out), "tce_mha_small64_f32: pointers must be 16-byte aligned");
  hipLaunchKernelGGL((mha_small_kernel<64, 128>), dim3(nheads, tce_cdiv(L, 32)), dim3(256), 0, (hipStream_t)stream, qkv, out, L,
                     nheads, scale, 1, (const float*)nullptr);
  TCE_CHECK_LAUNCH("tce_mha_small64_f32");
  return TCE_OK;
}

extern "C" int tce_mha_small64_splits_f32(const float* qkv_planes, int32_t splits, const float* bias, float* out, int32_t L,
                                          int32_t nheads, float scale, tceStream stream) {
  TCE_CHECK_ARG(qkv_planes && out && nheads > 0 && splits >= 1 && splits <= 64, "tce_mha_small64_splits_f32: bad arguments");
  TCE_CHECK_ARG(L > 0 && L <= 128, "tce_mha_small64_splits_f32: sequence length %d outside 1..128", L);
  TCE_CHECK_ARG(tce_aligned16(qkv_planes) && tce_aligned16(out) && (!bias || tce_aligned16(bias)),
                "tce_mha_small64_splits_f32: pointers must be 16-byte aligned");
  hipLaunchKernelGGL((mha_small_kernel<64, 128>), dim3(nheads, tce_cdiv(L, 32)), dim3(256), 0, (hipStream_t)stream, qkv_planes, out,
                     L, nheads, scale, splits, bias);
  TCE_CHECK_LAUNCH("tce_mha_small64_splits_f32");
  return TCE_OK;
}

extern "C" int tce_mha_small64_seqs_f32(const float* qkv_planes, int32_t splits, const float* bias, float* out, int32_t nseq, int32_t L,
                                        int32_t nheads, float scale, tceStream stream) {
  TCE_CHECK_ARG(qkv_planes && out && nheads > 0 && splits >= 1 && splits <= 64 && nseq > 0 && nseq <= 64, "tce_mha_small64_seqs_f32: bad arguments");
  TCE_CHECK_ARG(L > 0 && L <= 128, "tce_mha_small64_seqs_f32: sequence length %d outside 1..128", L);
  TCE_CHECK_ARG(tce_aligned16(qkv_planes) && tce_aligned16(out) && (!bias || tce_aligned16(bias)),
                "tce_mha_small64_seqs_f32: pointers must be 16-byte aligned");
  hipLaunchKernelGGL((mha_small_kernel<64, 128>), dim3(nheads, tce_cdiv(L, 32), nseq), dim3(256), 0, (hipStream_t)stream, qkv_planes,
                     out, L, nheads, scale, splits, bias);
  TCE_CHECK_LAUNCH("tce_mha_small64_seqs_f32");
  return TCE_OK;
}

extern "C" int tce_tanh_f32(const float* x, float* out, int64_t n, tceStream stream) {
  TCE_CHECK_ARG(x && out && n > 0, "tce_tanh_f32: bad arguments");
  hipLaunchKernelGGL(tanh_kernel, dim3(tce_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, (long long)n);
  TCE_CHECK_LAUNCH("tce_tanh_f32");
  return TCE_OK;
}
