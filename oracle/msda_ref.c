/* ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product path).
 *
 * Plain-C scalar restatement of the reference's multi-scale deformable attention forward rule
 * (/root/reference/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:34-85 bilinear, :421-452 2-D branch;
 *  host wrapper ms_deform_attn_cuda.cu:21-102): one output scalar per (b, q, m, c),
 *    out = sum_l sum_p w[b,q,m,l,p] * bilinear(value_l[b,:,m,c], loc[b,q,m,l,p])
 *    h_im = loc_y*H_l - 0.5, w_im = loc_x*W_l - 0.5; sample counted iff -1 < h_im < H_l and -1 < w_im < W_l;
 *    each corner is dropped individually when it lies outside the level (zero padding).
 * Pinned by tests/golden/msda_cases.npz (outputs of the reference's own ms_deform_attn_core_pytorch, incl. the
 * case of models/ops/test.py:21-26).  Build: oracle/build_oracle.py -> oracle/libmsda_ref.so
 */
#include <math.h>
#include <stdint.h>

static float bilinear(const float* v, int H, int W, int M, int D, float h, float w, int m, int c) {
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
  const long w_stride = (long)M * D, h_stride = (long)W * w_stride, base = (long)m * D + c;
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = v[h_low * h_stride + w_low * w_stride + base];
  if (h_low >= 0 && w_high <= W - 1) v2 = v[h_low * h_stride + w_high * w_stride + base];
  if (h_high <= H - 1 && w_low >= 0) v3 = v[h_high * h_stride + w_low * w_stride + base];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = v[h_high * h_stride + w_high * w_stride + base];
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

void msda_ref_forward(const float* value, const int64_t* shapes, const int64_t* level_start, const float* loc,
                      const float* weight, float* out, int N, int S, int M, int D, int Lq, int L, int P) {
  for (int b = 0; b < N; ++b)
    for (int q = 0; q < Lq; ++q)
      for (int m = 0; m < M; ++m)
        for (int c = 0; c < D; ++c) {
          float col = 0;
          for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const float* v = value + ((long)b * S + level_start[l]) * M * D;
            for (int p = 0; p < P; ++p) {
              const long i = ((((long)b * Lq + q) * M + m) * L + l) * P + p;
              const float h_im = loc[2 * i + 1] * H - 0.5f, w_im = loc[2 * i] * W - 0.5f;
              if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) col += bilinear(v, H, W, M, D, h_im, w_im, m, c) * weight[i];
            }
          }
          out[(((long)b * Lq + q) * M + m) * D + c] = col;
        }
}
