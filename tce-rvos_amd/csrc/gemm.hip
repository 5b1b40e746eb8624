// fp32 GEMM / implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C[M,N] = epi((A (+A2)) @ W^T + bias)      A [M,K] row-major, W [N,K] row-major (nn.Linear layout)
//
// Structure (one workgroup = 256 threads = 4 waves in a 2x2 grid over a BM x BN output tile):
//   * K is walked in BK = 16 slices.  Each thread stages float4 pieces of the A and W tiles global -> VGPR
//     (next slice prefetched while the current one is multiplied) -> LDS, stored K-MAJOR ([k][m], [k][n])
//     so that one MFMA operand fetch is 32 consecutive floats per half-wave (conflict-free ds_read_b32):
//     lane l supplies A[m = l&31][k = l>>5] and B[k = l>>5][n = l&31].
//   * two LDS buffers, one barrier per K slice.
//   * accumulators: (BM/64) x (BN/64) tiles of 32x32 per wave (f32x16 each); C/D map col = lane&31,
//     row = (r&3) + 8*(r>>2) + 4*(lane>>5): the N index is on the lanes, so every store is a 128-byte row piece.
//   * epilogue fused: bias, ReLU / GELU(erf), residual add or multiply.
//   * CONV: the A loader computes the source pixel of each output row for the current (ky,kx) tap
//     (zero outside the image) -- no im2col buffer.
//   * tile ids are remapped so that each XCD (private L2) owns a contiguous run of tiles.
// fp32 MFMA issues at the f32 vector rate (157 TFLOP/s peak); it is bit-exact f32 fmaf accumulation in k order.
#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/tce_rvos.h"

namespace {

constexpr int BK = 16;

// __launch_bounds__(256, 3): without the bound the compiler spread the 128x128 tile over 124 VGPRs + 64 AGPRs (two waves per SIMD);
// with it the same code fits 112 registers, no spills: FOUR waves per SIMD (33 KB of LDS per workgroup), which is what keeps the
// 64-cycle fp32 MFMA pipe fed across the per-slice barrier (round 5; the counters had shown it busy 67 % of the time).
template <int BM, int BN, bool CONV>
__global__ void __launch_bounds__(256, 3) gemm_f32_kernel(const tceGemmArgs p, const int tiles_m, const int tiles_n) {
  constexpr int LDAS = BM + 4;
  constexpr int LDBS = BN + 4;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NA = BM / 64;  // float4 per thread for the A tile
  constexpr int NB = BN / 64;
  __shared__ float smem[2 * BK * (LDAS + LDBS)];
  float* As0 = smem;
  float* Bs0 = smem + 2 * BK * LDAS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int bz = blockIdx.z;

  const float* __restrict__ A = p.A + (long long)bz * p.sA;
  const float* __restrict__ A2 = p.A2 ? p.A2 + (long long)bz * p.sA2 : nullptr;
  const float* __restrict__ W = p.W + (long long)bz * p.sW;
  const float* __restrict__ bias = p.bias ? p.bias + (long long)bz * p.sBias : nullptr;
  const float* __restrict__ res = p.res ? p.res + (long long)bz * p.sRes : nullptr;
  float* __restrict__ C = p.C + (long long)bz * p.sC;

  const int kq = tid & 3;      // which float4 of the 16-wide K slice
  const int lrow = tid >> 2;   // 0..63
  // --- per-thread row bookkeeping for the A loader
  long long a_off[NA];
  bool a_ok[NA];
  int c_t[NA], c_y[NA], c_x[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int gm = tm * BM + lrow + 64 * i;
    a_ok[i] = gm < p.M;
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int t = gm / hw, rem = gm - t * hw;
      c_t[i] = t;
      c_y[i] = (rem / p.Wo) * p.stride - p.pad;
      c_x[i] = (rem % p.Wo) * p.stride - p.pad;
      a_off[i] = 0;
    } else {
      a_off[i] = (long long)gm * p.lda;
      c_t[i] = c_y[i] = c_x[i] = 0;
    }
  }
  long long w_off[NB];
  bool w_ok[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int gn = tn * BN + lrow + 64 * i;
    w_ok[i] = gn < p.N;
    w_off[i] = (long long)gn * p.ldw;
  }

  f32x4 ra[NA], rb[NB];
  auto load_tiles = [&](int kt) {
    const int k0 = kt * BK + kq * 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (CONV) {
        const int tap = (kt * BK) / p.Cin;
        const int c0 = k0 - tap * p.Cin;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        const int yi = c_y[i] + ky, xi = c_x[i] + kx;
        if (a_ok[i] && yi >= 0 && yi < p.H && xi >= 0 && xi < p.Wd) {
          v = *reinterpret_cast<const f32x4*>(A + (((long long)c_t[i] * p.H + yi) * p.Wd + xi) * p.Cin + c0);
        }
      } else {
        if (a_ok[i]) {
          v = *reinterpret_cast<const f32x4*>(A + a_off[i] + k0);
          if (A2) {
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(A2 + (long long)(tm * BM + lrow + 64 * i) * p.lda2 + k0);
            v += v2;
          }
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (w_ok[i]) v = *reinterpret_cast<const f32x4*>(W + w_off[i] + k0);
      rb[i] = v;
    }
  };
  auto store_tiles = [&](int buf) {
    float* As = As0 + buf * BK * LDAS;
    float* Bs = Bs0 + buf * BK * LDBS;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int m = lrow + 64 * i;
#pragma unroll
      for (int j = 0; j < 4; ++j) As[(kq * 4 + j) * LDAS + m] = ra[i][j];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int n = lrow + 64 * i;
#pragma unroll
      for (int j = 0; j < 4; ++j) Bs[(kq * 4 + j) * LDBS + n] = rb[i][j];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  const int l31 = lane & 31, lhi = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);
    const float* As = As0 + buf * BK * LDAS + wm * WM + l31;
    const float* Bs = Bs0 + buf * BK * LDBS + wn * WN + l31;
    // operands of k-step kk+1 are fetched while the MFMAs of step kk issue (round 5: the counters showed the matrix pipe idle a
    // third of the time at two waves per SIMD -- every step waited for its own LDS reads; profiles/r05_mfma_util.txt)
    float a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = As[lhi * LDAS + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = Bs[lhi * LDBS + j * 32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      if (kk + 1 < BK / 2) {
        const int k = (kk + 1) * 2 + lhi;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[(kk + 1) & 1][i] = As[k * LDAS + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[(kk + 1) & 1][j] = Bs[k * LDBS + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[kk & 1][j], a[kk & 1][i], acc[i][j], 0, 0, 0);  // D[n][m]
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // --- epilogue
  const bool vec_ok = tce_epi_vec_ok(C, p.ldc, res, p.ldres, bias, p.res_mode);
  tce_amax_t amax = 0;  // unused: the exact-fp32 kernel has no operand-range contract
#define EPI_BODY(ACT, RES)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                                \
    const int row = tm * BM + wm * WM + i * 32 + l31;                                                             \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                \
        tce_epi_store_t<ACT, RES>(acc[i][j], bias, res, C, row, tn * BN + wn * WN + j * 32 + 4 * lhi, p.M, p.N,   \
                                  p.ldc, p.ldres, vec_ok, amax);                                                  \
  }
  TCE_EPI_DISPATCH(p.act, p.res_mode, EPI_BODY)
#undef EPI_BODY
}

template <int BM, int BN>
int launch(const tceGemmArgs& a, hipStream_t s) {
  const int tiles_m = tce_cdiv(a.M, BM), tiles_n = tce_cdiv(a.N, BN);
  dim3 grid(tiles_m * tiles_n, 1, a.batch > 0 ? a.batch : 1);
  if (a.conv)
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true>), grid, dim3(256), 0, s, a, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false>), grid, dim3(256), 0, s, a, tiles_m, tiles_n);
  return 0;
}

}  // namespace

int tce_gemm_f16x3_launch_big(const tceGemmArgs& a, int tile, hipStream_t s);    // gemm_f16x3_big.hip
int tce_gemm_f16x3_launch_small(const tceGemmArgs& a, int tile, hipStream_t s);  // gemm_f16x3_small.hip

// 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32); 1: 3 x fp16 split MFMA (fp32-accurate);
// 2: single fp16 MFMA per product, operands rounded to nearest fp16, fp32 accumulate (BASELINE config 5's arithmetic)
// The process default (tce_set_gemm_mode) and a per-THREAD override (tce_set_gemm_mode_thread; -1 = none): per-site
// arithmetic switches the mode dozens of times per clip around groups of launches, and the mode is read when a launch is
// ISSUED -- with one process-wide variable a second host thread issuing launches meanwhile would run (or capture) the
// wrong arithmetic silently.  A thread's switches are now its own.
static int g_gemm_mode_default = 1;
static thread_local int t_gemm_mode = -1;
#define g_gemm_mode (t_gemm_mode >= 0 ? t_gemm_mode : g_gemm_mode_default)
int tce_gemm_single_pass() { return g_gemm_mode == 2; }

extern "C" int tce_set_gemm_mode(int32_t mode) {
  TCE_CHECK_ARG(mode >= 0 && mode <= 2, "tce_set_gemm_mode: mode must be 0 (f32), 1 (3xf16 split) or 2 (single f16)");
  g_gemm_mode_default = mode;
  return TCE_OK;
}
extern "C" int tce_set_gemm_mode_thread(int32_t mode) {
  TCE_CHECK_ARG(mode >= -1 && mode <= 2, "tce_set_gemm_mode_thread: mode must be -1 (no override), 0, 1 or 2");
  t_gemm_mode = mode;
  return TCE_OK;
}
extern "C" int tce_get_gemm_mode(void) { return g_gemm_mode; }

// tile choice: the largest tile that still yields >= 2 workgroups per CU; fp32 MFMA is slow enough
// (64 cycles per 32x32x2) that the smaller tiles' extra LDS traffic is hidden, so favour grid fill.
static int g_force_tile = 0;
static bool g_rules_r4 = false;
extern "C" int tce_gemm_force_tile(int32_t tile) {  // 0 = automatic; 128128 / 12864 / 6464 pin the tile; -1 = automatic by round 4's rules (tuning aids)
  TCE_CHECK_ARG(tile == 0 || tile == -1 || tile == 256128 || tile == 128128 || tile == 12864 || tile == 6464 || tile == 6465,
                "tce_gemm_force_tile: bad tile (0, -1, 256128, 128128, 12864, 6464 or 6465)");
  g_force_tile = tile > 0 ? tile : 0;
  g_rules_r4 = tile == -1;
  return TCE_OK;
}

static int select_tile_ex(int M, int N, int K, int batch, int conv) {
  if (g_force_tile) return g_force_tile;
  const long long b = batch > 0 ? batch : 1;
  const long long n256 = (long long)tce_cdiv(M, 256) * tce_cdiv(N, 128) * b;
  const long long n128 = (long long)tce_cdiv(M, 128) * tce_cdiv(N, 128) * b;
  const long long n12864 = (long long)tce_cdiv(M, 128) * tce_cdiv(N, 64) * b;
  if (g_gemm_mode >= 1) {
    // measured on MI355X (tools/gemm_bench.py): the 8-wave 256x128 tile wins once it fills the chip 1.5x and the
    // problem is wide or deep; 128x128 for the big implicit-GEMM convolutions; the small tiles elsewhere
    // (tools/gemm_shape_sweep.py over every launch of a config-2 clip: deep-K / wide problems already prefer the
    // 8-wave tile at ~0.65 waves of the chip, e.g. 24100x256x2048 90 us vs 100 us, 4600x1536x384 31 us vs 37 us)
    if (!conv && ((n256 >= 384 && N >= 512) || (n256 >= 128 && K >= 1024) || (n256 >= 160 && N >= 1024 && K >= 384))) return 256128;
    // round 5, the shapes of an 8-clip group and of the 72000-row maps (tools/runs/r6r.sh, every tile forced): 1.5 rounds of the
    // chip are enough at N = 256 / 384 too (36800x384x384 48.5 vs 54.3 us, 192800x256x256 117 vs 150, 576000x256x96 212 vs 287),
    // and a square-ish deep problem prefers it from 0.9 rounds (9600x768x768 40.1 vs 52.8); un-batched launches only (what was measured)
    if (!conv && !g_rules_r4 && b == 1 && ((n256 >= 384 && N >= 256) || (n256 >= 224 && N >= 768 && K >= 768))) return 256128;
    if (conv && n128 >= 256) return 128128;
    if (n12864 >= 384 && M > 64) return 12864;
    // deep K with a moderate grid (Swin stage 3's fc2, 4600 x 384 x 1536: 216 tiles of 128x64): the wider tile halves the
    // operand conversions per MFMA over 48 K slices -- 33.7 vs 39.3 us (tools/gemm_shape_bench.py, ALL_TILES=1)
    if (n12864 >= 192 && K >= 1024 && M > 64) return 12864;
    return 6464;
  }
  if (n128 >= 512 && N > 64) return 128128;
  if (n12864 >= 384 && M > 64) return 12864;
  return 6464;
}

extern "C" int tce_gemm_select_tile(int32_t M, int32_t N, int32_t batch) { return select_tile_ex(M, N, 0, batch, 0); }
extern "C" int tce_gemm_select_tile_ex(int32_t M, int32_t N, int32_t K, int32_t batch, int32_t conv) {
  return select_tile_ex(M, N, K, batch, conv);
}

namespace {
__global__ void __launch_bounds__(256) epi_fix_kernel(float* __restrict__ C, const float* __restrict__ res, const int M,
                                                      const int N, const long long ldc, const long long ldres,
                                                      const long long sC, const long long sRes, const int res_mode,
                                                      const int relu) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  float* c = C + (long long)blockIdx.y * sC + (long long)m * ldc + n;
  float v = *c;
  if (res_mode) {
    const float r = res[(long long)blockIdx.y * sRes + (long long)m * ldres + n];
    v = res_mode == 1 ? v + r : v * r;
  }
  if (relu) v = fmaxf(v, 0.f);
  *c = v;
}
}  // namespace

static int launch_epi_fix(const tceGemmArgs& a, int res_mode, bool relu, hipStream_t s) {
  const long long total = (long long)a.M * a.N;
  hipLaunchKernelGGL(epi_fix_kernel, dim3(tce_cdiv(total, 256), a.batch > 0 ? a.batch : 1), dim3(256), 0, s, a.C, a.res, a.M, a.N,
                     (long long)a.ldc, (long long)a.ldres, (long long)a.sC, (long long)a.sRes, res_mode, relu ? 1 : 0);
  TCE_CHECK_LAUNCH("tce_gemm_f32(epilogue fix-up)");
  return TCE_OK;
}

extern "C" int tce_gemm_f32(const tceGemmArgs* args, tceStream stream) {
  TCE_CHECK_ARG(args != nullptr, "tce_gemm_f32: null args");
  tceGemmArgs a = *args;
  TCE_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "tce_gemm_f32: M,N,K must be positive (got %d,%d,%d)", a.M, a.N, a.K);
  TCE_CHECK_ARG(a.K % BK == 0, "tce_gemm_f32: K=%d must be a multiple of %d", a.K, BK);
  TCE_CHECK_ARG(a.A && a.W && a.C, "tce_gemm_f32: null A/W/C");
  TCE_CHECK_ARG(tce_aligned16(a.A) && tce_aligned16(a.W) && (!a.A2 || tce_aligned16(a.A2)),
                "tce_gemm_f32: A/A2/W must be 16-byte aligned");
  TCE_CHECK_ARG(a.ldw % 4 == 0 && a.ldw >= a.K, "tce_gemm_f32: ldw=%d must be >= K and a multiple of 4", a.ldw);
  TCE_CHECK_ARG(a.ldc >= a.N, "tce_gemm_f32: ldc=%d < N=%d", a.ldc, a.N);
  TCE_CHECK_ARG(a.res_mode == 0 || (a.res && a.ldres >= a.N), "tce_gemm_f32: res_mode set but res/ldres invalid");
  TCE_CHECK_ARG(a.act >= 0 && a.act <= 3 && a.res_mode >= 0 && a.res_mode <= 2, "tce_gemm_f32: bad act/res_mode");
  if (a.batch <= 0) a.batch = 1;
  if (a.conv) {
    TCE_CHECK_ARG(a.A2 == nullptr, "tce_gemm_f32: A2 is not supported with conv");
    TCE_CHECK_ARG(a.Cin % BK == 0, "tce_gemm_f32: conv Cin=%d must be a multiple of %d", a.Cin, BK);
    // batch > 1 with conv = split-K chunks of one convolution (only issued by tce_gemm_splitk_f32, split-fp16 mode)
    TCE_CHECK_ARG((long long)a.K * a.batch == (long long)a.kh * a.kw * a.Cin, "tce_gemm_f32: conv K=%d x batch != kh*kw*Cin", a.K);
    TCE_CHECK_ARG(a.batch == 1 || g_gemm_mode >= 1, "tce_gemm_f32: split convolutions need an fp16-MFMA mode");
    TCE_CHECK_ARG(a.M == a.T * a.Ho * a.Wo, "tce_gemm_f32: conv M=%d != T*Ho*Wo", a.M);
    TCE_CHECK_ARG(a.Ho == (a.H + 2 * a.pad - a.kh) / a.stride + 1 && a.Wo == (a.Wd + 2 * a.pad - a.kw) / a.stride + 1,
                  "tce_gemm_f32: conv output size mismatch");
  } else {
    TCE_CHECK_ARG(a.lda % 4 == 0 && a.lda >= a.K, "tce_gemm_f32: lda=%d must be >= K and a multiple of 4", a.lda);
    TCE_CHECK_ARG(!a.A2 || (a.lda2 % 4 == 0 && a.lda2 >= a.K), "tce_gemm_f32: lda2 invalid");
  }
  hipStream_t s = (hipStream_t)stream;
  // epilogue combinations without a specialised kernel body (gemm_epilogue.h) = GEMM with the activation only, then one
  // elementwise pass for the residual (and a trailing ReLU for act 3)
  bool fix = false, fix_relu = false;
  int fix_res_mode = 0;
  if (a.act == 3 && a.res_mode == 0) a.act = 1;
  if (!tce_epi_supported(a.act, a.res_mode)) {
    TCE_CHECK_ARG(a.res != a.C, "tce_gemm_f32: act=%d with res_mode=%d has no fused epilogue; it cannot run in place", a.act,
                  a.res_mode);
    fix = true;
    fix_res_mode = a.res_mode;
    fix_relu = a.act == 3;
    if (a.act == 3) a.act = 0;
    a.res_mode = 0;
  }
  const int tile = select_tile_ex(a.M, a.N, a.K, a.batch, a.conv);
  if (g_gemm_mode >= 1 && a.K % 32 == 0 && (!a.conv || a.Cin % 32 == 0)) {
    if (tile == 256128 || tile == 128128) tce_gemm_f16x3_launch_big(a, tile, s);
    else tce_gemm_f16x3_launch_small(a, tile, s);
    TCE_CHECK_LAUNCH("tce_gemm_f32(f16x3)");
    return fix ? launch_epi_fix(a, fix_res_mode, fix_relu, s) : TCE_OK;
  }
  if (tile == 128128 || tile == 256128) launch<128, 128>(a, s);
  else if (tile == 6465) launch<64, 64>(a, s);
  else if (tile == 12864) launch<128, 64>(a, s);
  else launch<64, 64>(a, s);
  TCE_CHECK_LAUNCH("tce_gemm_f32");
  return fix ? launch_epi_fix(a, fix_res_mode, fix_relu, s) : TCE_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Split-K for skinny, deep problems (M <= a tile or two, K in the thousands: the level-3 3x3/s2 convolution of C5 with
// 300 output rows and K = 6912, the RoBERTa projections at 32 tokens, the decoder FFNs on 25 rows): a plain launch has N/64 workgroups streaming the whole K extent of the weights one
// after the other (9.4 MB at ~0.2 TB/s for 32x768x3072).  The K extent is cut into `splits` chunks that run as the
// batch dimension of the same kernel into workspace[splits][M][N]; a second kernel sums the partials and applies the
// epilogue (bias, activation, residual) exactly as the GEMM epilogue would.
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                            const float* __restrict__ res, float* __restrict__ C,
                                                            const int M, const int N, const int splits, const int ldc,
                                                            const int ldres, const int act, const int res_mode) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = N >> 2;
  if (i >= (long long)M * n4) return;
  const int m = (int)(i / n4), n = (int)(i % n4) * 4;
  const long long plane = (long long)M * N;
  f32x4 v = *reinterpret_cast<const f32x4*>(ws + (long long)m * N + n);
  for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4*>(ws + s * plane + (long long)m * N + n);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float x = v[c] + (bias ? bias[n + c] : 0.f);
    if (act == 1) x = fmaxf(x, 0.f);
    if (act == 2) x = tce_gelu(x);
    if (res_mode == 1) x += res[(long long)m * ldres + n + c];
    if (res_mode == 2) x *= res[(long long)m * ldres + n + c];
    if (act == 3) x = fmaxf(x, 0.f);
    C[(long long)m * ldc + n + c] = x;
  }
}
// The same reduction with the LayerNorm that follows it in every post-norm block (RoBERTa's attention.output / output
// sub-layers, the decoder's FFN): one wavefront per output row (N <= 1024) sums the splits, adds bias + residual and
// normalises the row -- the LayerNorm launch behind a split-K GEMM disappears (24 per clip in the text encoder alone).
// Few-row form of the reduction + LayerNorm (M <= 256 rows: RoBERTa at 32 tokens, the decoder at 25): ONE WORKGROUP per row,
// thread t owns float4 t of the row (N <= 1024), so every partial plane's load of the row is in flight at once -- one memory
// round trip instead of the wave-per-row kernel's serial walk over the planes (8.6 us for 32 x 768 x 16 planes on 8
// workgroups).  Same arithmetic: planes summed in split order, + bias + residual, two-pass mean / variance over the row.
__global__ void __launch_bounds__(256) splitk_reduce_ln_row_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                                   const float* res, float* C, const int M, const int N,
                                                                   const int splits, const int ldc, const int ldres,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   const float eps) {
  __shared__ float sPart[2][4];
  const int m = blockIdx.x, t = threadIdx.x, n4 = N >> 2;
  const long long plane = (long long)M * N;
  const bool on = t < n4;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (on) {
    const float* p = ws + (long long)m * N + 4 * t;
    a = *reinterpret_cast<const f32x4*>(p);
    for (int s = 1; s < splits; ++s) a += *reinterpret_cast<const f32x4*>(p + (long long)s * plane);
    if (bias) a += *reinterpret_cast<const f32x4*>(bias + 4 * t);
    if (res) a += *reinterpret_cast<const f32x4*>(res + (long long)m * ldres + 4 * t);
  }
  float sum = wave_sum(on ? (a[0] + a[1]) + (a[2] + a[3]) : 0.f);
  if ((t & 63) == 0) sPart[0][t >> 6] = sum;
  __syncthreads();
  const float mean = ((sPart[0][0] + sPart[0][1]) + (sPart[0][2] + sPart[0][3])) / (float)N;
  float sq = 0.f;
  if (on)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = a[c] - mean;
      sq = fmaf(d, d, sq);
    }
  sq = wave_sum(sq);
  if ((t & 63) == 0) sPart[1][t >> 6] = sq;
  __syncthreads();
  const float rstd = rsqrtf(((sPart[1][0] + sPart[1][1]) + (sPart[1][2] + sPart[1][3])) / (float)N + eps);
  if (on) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * t), b = *reinterpret_cast<const f32x4*>(beta + 4 * t);
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = (a[c] - mean) * rstd * g[c] + b[c];
    *reinterpret_cast<f32x4*>(C + (long long)m * ldc + 4 * t) = o;
  }
}

// res may alias C (every caller runs it in place on the residual stream): neither is __restrict__, and a row's residual
// is fully read (it feeds the mean) before the row's first store -- one wavefront owns the row.
__global__ void __launch_bounds__(256) splitk_reduce_ln_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                               const float* res, float* C,
                                                               const int M, const int N, const int splits, const int ldc,
                                                               const int ldres, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float eps) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const long long plane = (long long)M * N;
  const int n4 = N >> 2;  // float4 per row, <= 256: lane takes pieces lane, lane + 64, ...
  f32x4 v[4];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = lane + 64 * j;
    v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q < n4) {
      const float* p = ws + (long long)m * N + 4 * q;
      // partial sums in split order (deterministic); four loads in flight per step
      f32x4 a = *reinterpret_cast<const f32x4*>(p);
      int s = 1;
      for (; s + 3 < splits; s += 4) {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(p + (long long)s * plane);
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(p + (long long)(s + 1) * plane);
        const f32x4 p2 = *reinterpret_cast<const f32x4*>(p + (long long)(s + 2) * plane);
        const f32x4 p3 = *reinterpret_cast<const f32x4*>(p + (long long)(s + 3) * plane);
        a += p0;
        a += p1;
        a += p2;
        a += p3;
      }
      for (; s < splits; ++s) a += *reinterpret_cast<const f32x4*>(p + (long long)s * plane);
      if (bias) a += *reinterpret_cast<const f32x4*>(bias + 4 * q);
      if (res) a += *reinterpret_cast<const f32x4*>(res + (long long)m * ldres + 4 * q);
      v[j] = a;
      sum += (a[0] + a[1]) + (a[2] + a[3]);
    }
  }
  const float mean = wave_sum(sum) / (float)N;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (lane + 64 * j < n4)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = v[j][c] - mean;
        sq = fmaf(d, d, sq);
      }
  const float rstd = rsqrtf(wave_sum(sq) / (float)N + eps);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = lane + 64 * j;
    if (q < n4) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * q), b = *reinterpret_cast<const f32x4*>(beta + 4 * q);
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = (v[j][c] - mean) * rstd * g[c] + b[c];
      *reinterpret_cast<f32x4*>(C + (long long)m * ldc + 4 * q) = o;
    }
  }
}
}  // namespace

static int splitk_impl(const tceGemmArgs* args, int32_t splits, float* workspace, const float* gamma, const float* beta, float eps,
                       tceStream stream);

extern "C" int tce_gemm_splitk_f32(const tceGemmArgs* args, int32_t splits, float* workspace, tceStream stream) {
  return splitk_impl(args, splits, workspace, nullptr, nullptr, 0.f, stream);
}

// The reduction pass alone: C = LN?( epi( sum_s ws[s] + bias ) ) for partial planes produced by tce_thin_partials_f32
extern "C" int tce_splitk_reduce_f32(const float* ws, int32_t splits, int32_t M, int32_t N, const float* bias, int32_t act,
                                     const float* res, int32_t ldres, int32_t res_mode, float* C, int32_t ldc, const float* gamma,
                                     const float* beta, float eps, tceStream stream) {
  TCE_CHECK_ARG(ws && C && splits >= 1 && splits <= 64 && M > 0 && N > 0 && N % 4 == 0 && ldc >= N, "tce_splitk_reduce_f32: bad arguments");
  TCE_CHECK_ARG(act >= 0 && act <= 3 && res_mode >= 0 && res_mode <= 2 && (res_mode == 0 || (res && ldres >= N)),
                "tce_splitk_reduce_f32: bad act / res_mode");
  TCE_CHECK_ARG(tce_aligned16(ws) && (!bias || tce_aligned16(bias)), "tce_splitk_reduce_f32: ws / bias must be 16-byte aligned");
  if (gamma) {
    TCE_CHECK_ARG(beta && act == 0 && res_mode != 2 && N <= 1024 && ldc % 4 == 0 && (res_mode == 0 || ldres % 4 == 0) &&
                      tce_aligned16(C) && tce_aligned16(gamma) && tce_aligned16(beta) && (res_mode == 0 || tce_aligned16(res)),
                  "tce_splitk_reduce_f32: LayerNorm form: no activation, additive residual, N <= 1024, 16-byte aligned rows");
    if (M <= 256)
      hipLaunchKernelGGL(splitk_reduce_ln_row_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, ws, bias,
                         res_mode == 1 ? res : nullptr, C, M, N, splits, ldc, ldres, gamma, beta, eps);
    else
      hipLaunchKernelGGL(splitk_reduce_ln_kernel, dim3(tce_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, ws, bias,
                         res_mode == 1 ? res : nullptr, C, M, N, splits, ldc, ldres, gamma, beta, eps);
  } else {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(tce_cdiv((long long)M * (N / 4), 256)), dim3(256), 0, (hipStream_t)stream, ws, bias,
                       res, C, M, N, splits, ldc, ldres, act, res_mode);
  }
  TCE_CHECK_LAUNCH("tce_splitk_reduce_f32");
  return TCE_OK;
}

extern "C" int tce_gemm_splitk_ln_f32(const tceGemmArgs* args, int32_t splits, float* workspace, const float* gamma,
                                      const float* beta, float eps, tceStream stream) {
  TCE_CHECK_ARG(args && gamma && beta, "tce_gemm_splitk_ln_f32: null pointer");
  TCE_CHECK_ARG(args->act == 0 && args->res_mode != 2 && args->N <= 1024 && args->N % 4 == 0 && !args->conv,
                "tce_gemm_splitk_ln_f32: plain GEMM, no activation, additive residual, N <= 1024");
  TCE_CHECK_ARG(args->ldc % 4 == 0 && (args->res_mode == 0 || args->ldres % 4 == 0) && tce_aligned16(args->C) &&
                    tce_aligned16(gamma) && tce_aligned16(beta) && (!args->bias || tce_aligned16(args->bias)) &&
                    (args->res_mode == 0 || tce_aligned16(args->res)),
                "tce_gemm_splitk_ln_f32: C / res / bias / gamma / beta must be 16-byte aligned with pitches of 4 floats");
  return splitk_impl(args, splits, workspace, gamma, beta, eps, stream);
}

static int splitk_impl(const tceGemmArgs* args, int32_t splits, float* workspace, const float* gamma, const float* beta, float eps,
                       tceStream stream) {
  TCE_CHECK_ARG(args != nullptr && workspace != nullptr, "tce_gemm_splitk_f32: null args/workspace");
  tceGemmArgs a = *args;
  TCE_CHECK_ARG(splits >= 1 && splits <= 64, "tce_gemm_splitk_f32: splits=%d out of range", splits);
  TCE_CHECK_ARG(a.batch <= 1, "tce_gemm_splitk_f32: un-batched problems only");
  if (a.conv && (g_gemm_mode < 1 || a.Cin % 32 != 0)) splits = 1;  // the exact-fp32 kernel has no split convolution
  TCE_CHECK_ARG(a.M > 0 && a.N > 0 && a.N % 4 == 0 && a.K > 0 && a.K % (splits * BK) == 0,
                "tce_gemm_splitk_f32: N=%d must be a multiple of 4 and K=%d a multiple of splits*%d", a.N, a.K, BK);
  TCE_CHECK_ARG(a.C && tce_aligned16(workspace), "tce_gemm_splitk_f32: null C / unaligned workspace");
  TCE_CHECK_ARG(a.ldc >= a.N && (a.res_mode == 0 || (a.res && a.ldres >= a.N)), "tce_gemm_splitk_f32: bad ldc/res");
  TCE_CHECK_ARG(a.act >= 0 && a.act <= 3 && a.res_mode >= 0 && a.res_mode <= 2, "tce_gemm_splitk_f32: bad act/res_mode");
  tceGemmArgs g = a;
  const int kc = a.K / splits;
  g.K = kc;
  g.batch = splits;
  g.sA = a.conv ? 0 : kc; g.sA2 = kc; g.sW = kc; g.sBias = 0; g.sRes = 0;
  g.sC = (long long)a.M * a.N;
  g.C = workspace; g.ldc = a.N;
  g.bias = nullptr; g.res = nullptr; g.act = 0; g.res_mode = 0;
  const int st = tce_gemm_f32(&g, stream);
  if (st != TCE_OK) return st;
  const long long total = (long long)a.M * (a.N / 4);
  if (gamma && a.M <= 256)
    hipLaunchKernelGGL(splitk_reduce_ln_row_kernel, dim3(a.M), dim3(256), 0, (hipStream_t)stream, workspace, a.bias,
                       a.res_mode == 1 ? a.res : nullptr, a.C, a.M, a.N, splits, a.ldc, a.ldres, gamma, beta, eps);
  else if (gamma)
    hipLaunchKernelGGL(splitk_reduce_ln_kernel, dim3(tce_cdiv(a.M, 4)), dim3(256), 0, (hipStream_t)stream, workspace, a.bias,
                       a.res_mode == 1 ? a.res : nullptr, a.C, a.M, a.N, splits, a.ldc, a.ldres, gamma, beta, eps);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(tce_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, workspace, a.bias,
                       a.res, a.C, a.M, a.N, splits, a.ldc, a.ldres, a.act, a.res_mode);
  TCE_CHECK_LAUNCH("tce_gemm_splitk_f32(reduce)");
  return TCE_OK;
}
