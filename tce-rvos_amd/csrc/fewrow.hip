// Few-row linear layers (a few dozen rows): the frame-token path of the encoder (T*F = 40 rows at config 2,
// tce_deformable_transformer.py:443-484), the decoder's per-query projections (T*Q = 25 rows, :665-790) and the text-side
// key / value projections (<= 32 rows, segmentation.py:366-371).  On the tiled matrix-core GEMM such a launch is one
// pipeline fill long (7-12 us for 5 MFLOP) and each LayerNorm / sigmoid / second projection of the same rows is another
// launch; these sit on the clip's critical path between two encoder layers (0.54 ms per clip by ablation,
// profiles/r03_ablate_cfg2.txt).
//
//   out_s[r, n] = act_s( sum_k (x[r,k] (+ a2[r % a2_rows, k] if the segment asks)) W_s[n,k] + b_s[n] )      s < nseg <= 3
//   optional, segment 0 only:  out_0 += res   (the residual stream, in place)
//
// Exact fp32 on the vector ALUs (fmaf chains in k order: no operand splitting, no range contract).  One launch serves up to
// three projections of the SAME rows (q|k with the position map added and v without; reference points through a sigmoid
// beside the offsets|weights projection).  A workgroup owns an 8-column slab of one segment and 32 rows: x rows (+ addend)
// and the W slab are staged in LDS with full-line loads, thread (column c, row group g) accumulates row g with broadcast
// reads of x.  More slabs than CUs are never needed here (N <= 1024), so the launch is one wave of workgroups and costs
// about one memory round trip.
// NOT here: a LayerNorm epilogue.  It was built (every slab workgroup publishes its rows, the last arriver normalises them)
// and measured: the agent-scope release / acquire it needs is an L2 write-back + invalidate on this 8-XCD part, 17-50 us --
// more than the LayerNorm launch it saves (profiles/r03_fewrow.txt); removed.
#include "common.h"
#include "../../include/tce_rvos.h"

namespace {

constexpr int FR_COLS = 8;     // columns per slab
constexpr int FR_KC = 256;     // K chunk staged at a time
constexpr int FR_PITCH = FR_KC + 4;

struct FrSegDev {
  const float* W;
  const float* bias;
  float* out;
  int N, ldw, ldo, use_a2, act, slab0;  // slab0: first slab index of this segment
};

struct FrArgs {
  const float* x;
  const float* a2;
  const float* res;
  long long ldx, lda2, ldres;
  int a2_rows, R, K, nseg;
  FrSegDev seg[3];
  const float *g_in, *be_in;  // optional LayerNorm prologue over K (K <= FR_KC: the whole row is staged at once)
  float* xn_out;              // optional copy of the normalised rows (written by the workgroups of slab 0)
  long long ldxn;
  float eps_in;
};

// Thread (column c = tid & 7, row group g = tid >> 3) owns row g of the workgroup's 32-row pass (RPT = 1).  Rows past the
// end are staged as zeros, so the inner loop has no branches and its LDS reads can be issued ahead of the FMAs (a first
// version with per-row guards ran one LDS round trip per FMA group: 13 us for a 25 x 256 x 4 problem).
// A 64-row-per-workgroup variant (RPT = 2) was built and REMOVED: replayed inside the clip's hipGraph beside the pixel
// decoder's lateral branch it made the frame-token state differ from run to run (the replay-equals-eager tests caught it; 32-row
// passes never did in 35 replays); not reproduced at kernel level (tools/concurrency_probe.py), cause not established.
template <int RPT>
__global__ void __launch_bounds__(256) fewrow_linear_kernel(const FrArgs p) {
  __shared__ __attribute__((aligned(16))) float sX[32 * RPT * FR_PITCH];
  __shared__ __attribute__((aligned(16))) float sW[FR_COLS * FR_PITCH];
  const int tid = threadIdx.x;
  int s = 0;
  if (p.nseg > 1 && (int)blockIdx.x >= p.seg[1].slab0) s = 1;
  if (p.nseg > 2 && (int)blockIdx.x >= p.seg[2].slab0) s = 2;
  const FrSegDev sg = s == 0 ? p.seg[0] : (s == 1 ? p.seg[1] : p.seg[2]);
  const int n0 = ((int)blockIdx.x - sg.slab0) * FR_COLS;
  const int r0 = blockIdx.y * (32 * RPT);
  const int nr = min(32 * RPT, p.R - r0);
  const int c = tid & 7, g = tid >> 3;
  float acc[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < p.K; k0 += FR_KC) {
    const int kc = min(FR_KC, p.K - k0);  // multiple of 4
    const int q4 = kc >> 2;               // float4 per row (<= 64)
    if (k0) __syncthreads();
    // x rows (+ addend), zero rows past the end: thread t moves piece (t & 63) of rows (t >> 6), +4, ...
    if ((tid & 63) < q4) {
      const int q = tid & 63;
      for (int r = tid >> 6; r < 32 * RPT; r += 4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < nr) {
          v = *reinterpret_cast<const f32x4*>(p.x + (long long)(r0 + r) * p.ldx + k0 + 4 * q);
          if (p.g_in) {
            // LayerNorm prologue: a wavefront holds the whole row (K = 4 * q4 <= 256: one float4 per lane), two-pass statistics
            // like layernorm_kernel; `r` and `nr` are wave-uniform, so the shuffles run with all lanes of the wave
            const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) / (float)p.K;
            const f32x4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
            const float rstd = rsqrtf(wave_sum(fmaf(d[0], d[0], fmaf(d[1], d[1], fmaf(d[2], d[2], d[3] * d[3])))) / (float)p.K + p.eps_in);
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.g_in + 4 * q), b = *reinterpret_cast<const f32x4*>(p.be_in + 4 * q);
            v = f32x4{d[0] * rstd * g[0] + b[0], d[1] * rstd * g[1] + b[1], d[2] * rstd * g[2] + b[2], d[3] * rstd * g[3] + b[3]};
            if (p.xn_out && blockIdx.x == 0) *reinterpret_cast<f32x4*>(p.xn_out + (long long)(r0 + r) * p.ldxn + 4 * q) = v;
          }
          if (sg.use_a2) {
            const int ra = p.a2_rows > 0 ? (r0 + r) % p.a2_rows : r0 + r;
            v += *reinterpret_cast<const f32x4*>(p.a2 + (long long)ra * p.lda2 + k0 + 4 * q);
          }
        }
        *reinterpret_cast<f32x4*>(&sX[r * FR_PITCH + 4 * q]) = v;
      }
      for (int r = tid >> 6; r < FR_COLS; r += 4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n0 + r < sg.N) v = *reinterpret_cast<const f32x4*>(sg.W + (long long)(n0 + r) * sg.ldw + k0 + 4 * q);
        *reinterpret_cast<f32x4*>(&sW[r * FR_PITCH + 4 * q]) = v;
      }
    }
    __syncthreads();
    const float* wrow = &sW[c * FR_PITCH];
    const float* xrow = &sX[g * FR_PITCH];
    if (q4 == 64) {
#pragma unroll 8
      for (int q = 0; q < 64; ++q) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + 4 * q);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + i * 32 * FR_PITCH + 4 * q);
          acc[i] = fmaf(xv[0], wv[0], acc[i]);
          acc[i] = fmaf(xv[1], wv[1], acc[i]);
          acc[i] = fmaf(xv[2], wv[2], acc[i]);
          acc[i] = fmaf(xv[3], wv[3], acc[i]);
        }
      }
    } else {
      for (int q = 0; q < q4; ++q) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + 4 * q);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + i * 32 * FR_PITCH + 4 * q);
          acc[i] = fmaf(xv[0], wv[0], acc[i]);
          acc[i] = fmaf(xv[1], wv[1], acc[i]);
          acc[i] = fmaf(xv[2], wv[2], acc[i]);
          acc[i] = fmaf(xv[3], wv[3], acc[i]);
        }
      }
    }
  }
  const int n = n0 + c;
  if (n < sg.N) {
    const float bv = sg.bias ? sg.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = g + 32 * i;
      if (r < nr) {
        float v = acc[i] + bv;
        if (sg.act == 1) v = fmaxf(v, 0.f);
        if (sg.act == 2) v = 1.0f / (1.0f + __expf(-v));
        if (sg.act == 3) v = tce_gelu(v);
        if (s == 0 && p.res) v += p.res[(long long)(r0 + r) * p.ldres + n];
        sg.out[(long long)(r0 + r) * sg.ldo + n] = v;
      }
    }
  }
}

}  // namespace

extern "C" int tce_fewrow_linear_f32(const tceFewRowArgs* a, tceStream stream) {
  TCE_CHECK_ARG(a && a->x && a->nseg >= 1 && a->nseg <= 3, "tce_fewrow_linear_f32: bad arguments");
  TCE_CHECK_ARG(a->R > 0 && a->K > 0 && a->K % 4 == 0 && a->ldx % 4 == 0 && a->ldx >= a->K && tce_aligned16(a->x),
                "tce_fewrow_linear_f32: K and ldx must be multiples of 4, ldx >= K, x 16-byte aligned");
  // [begin, end) of a row-strided operand in floats
  auto span = [&](const float* p, long long ld, long long cols, long long rows, const float*& b, const float*& e) {
    b = p;
    e = p + (rows - 1) * ld + cols;
  };
  auto disjoint = [](const float* b0, const float* e0, const float* b1, const float* e1) { return e0 <= b1 || e1 <= b0; };
  FrArgs p;
  p.g_in = a->g_in; p.be_in = a->be_in; p.xn_out = a->g_in ? a->xn_out : nullptr; p.ldxn = a->ldxn; p.eps_in = a->eps_in;
  if (a->g_in) {
    // the prologue needs the whole row in one staged chunk with every lane of a wave on it (lanes >= K/4 would sit out the shuffles)
    TCE_CHECK_ARG(a->be_in && a->K == FR_KC && tce_aligned16(a->g_in) && tce_aligned16(a->be_in),
                  "tce_fewrow_linear_f32: the LayerNorm prologue needs K = %d and 16-byte aligned gamma / beta", FR_KC);
    TCE_CHECK_ARG(!a->xn_out || (a->ldxn >= a->K && a->ldxn % 4 == 0 && tce_aligned16(a->xn_out)), "tce_fewrow_linear_f32: xn_out pitch / alignment");
  }
  p.x = a->x; p.a2 = a->a2; p.res = a->res;
  p.ldx = a->ldx; p.lda2 = a->lda2; p.ldres = a->ldres;
  p.a2_rows = a->a2_rows; p.R = a->R; p.K = a->K; p.nseg = a->nseg;
  int slabs = 0;
  for (int s = 0; s < 3; ++s) {
    FrSegDev& d = p.seg[s];
    d = FrSegDev{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 1 << 30};
    if (s >= a->nseg) continue;
    const tceFewRowSeg& g = a->seg[s];
    TCE_CHECK_ARG(g.W && g.out && g.N > 0 && g.ldw % 4 == 0 && g.ldw >= a->K && g.ldo >= g.N && tce_aligned16(g.W),
                  "tce_fewrow_linear_f32: segment %d: bad weight / output geometry", s);
    TCE_CHECK_ARG(!g.use_a2 || (a->a2 && a->lda2 % 4 == 0 && tce_aligned16(a->a2)), "tce_fewrow_linear_f32: addend missing / misaligned");
    TCE_CHECK_ARG(g.act >= 0 && g.act <= 3, "tce_fewrow_linear_f32: act must be 0 (none), 1 (ReLU), 2 (sigmoid) or 3 (GELU)");
    {  // every workgroup re-reads the x (and addend) rows while others store: an output must not overlap them, nor another
       // segment's output; the residual may only BE segment 0's output (in place), never overlap anything else that is written
      const float *xb, *xe, *ob, *oe;
      span(a->x, a->ldx, a->K, a->R, xb, xe);
      span(g.out, g.ldo, g.N, a->R, ob, oe);
      TCE_CHECK_ARG(disjoint(xb, xe, ob, oe), "tce_fewrow_linear_f32: segment %d: out overlaps x", s);
      if (a->a2) {
        const float *ab, *ae;
        span(a->a2, a->lda2, a->K, a->a2_rows > 0 ? a->a2_rows : a->R, ab, ae);
        TCE_CHECK_ARG(disjoint(ab, ae, ob, oe), "tce_fewrow_linear_f32: segment %d: out overlaps a2", s);
      }
      for (int t = 0; t < s; ++t) {
        const float *pb, *pe;
        span(a->seg[t].out, a->seg[t].ldo, a->seg[t].N, a->R, pb, pe);
        TCE_CHECK_ARG(disjoint(pb, pe, ob, oe), "tce_fewrow_linear_f32: outputs of segments %d and %d overlap", t, s);
      }
      if (a->g_in && a->xn_out) {  // the normalised copy is written while other workgroups read x / a2 and write their slabs
        const float *nb, *ne;
        span(a->xn_out, a->ldxn, a->K, a->R, nb, ne);
        TCE_CHECK_ARG(disjoint(nb, ne, xb, xe) && disjoint(nb, ne, ob, oe), "tce_fewrow_linear_f32: xn_out overlaps x or segment %d's out", s);
        if (a->a2) {
          const float *ab, *ae;
          span(a->a2, a->lda2, a->K, a->a2_rows > 0 ? a->a2_rows : a->R, ab, ae);
          TCE_CHECK_ARG(disjoint(nb, ne, ab, ae), "tce_fewrow_linear_f32: xn_out overlaps a2");
        }
      }
      if (a->res) {
        const float *rb, *re;
        span(a->res, a->ldres, a->seg[0].N, a->R, rb, re);
        const bool in_place = s == 0 && a->res == g.out && a->ldres == g.ldo;
        TCE_CHECK_ARG(in_place || disjoint(rb, re, ob, oe),
                      "tce_fewrow_linear_f32: segment %d: res overlaps out (only segment 0 may alias it, exactly)", s);
      }
    }
    d.W = g.W; d.bias = g.bias; d.out = g.out; d.N = g.N; d.ldw = g.ldw; d.ldo = g.ldo; d.use_a2 = g.use_a2; d.act = g.act;
    d.slab0 = slabs;
    slabs += tce_cdiv(g.N, FR_COLS);
  }
  TCE_CHECK_ARG(!a->res || a->ldres >= a->seg[0].N, "tce_fewrow_linear_f32: residual pitch");
#ifdef FEWROW_RPT2
  // The removed 64-row-per-workgroup dispatch of the round-3 incident, kept compilable for audit only:
  //   HIPCC_EXTRA=-DFEWROW_RPT2 python -m tce_rvos_amd.build --force.  Never part of the shipped library.
  if (a->R > 32 && a->R <= 64) {
    hipLaunchKernelGGL(fewrow_linear_kernel<2>, dim3(slabs, 1), dim3(256), 0, (hipStream_t)stream, p);
    TCE_CHECK_LAUNCH("tce_fewrow_linear_f32");
    return TCE_OK;
  }
#endif
  hipLaunchKernelGGL(fewrow_linear_kernel<1>, dim3(slabs, tce_cdiv(a->R, 32)), dim3(256), 0, (hipStream_t)stream, p);
  TCE_CHECK_LAUNCH("tce_fewrow_linear_f32");
  return TCE_OK;
}
