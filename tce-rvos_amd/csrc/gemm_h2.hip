// PROTOTYPE (tools/gemm_h2_bench.py, tools/gemm_h2_stamps.py, tools/mfma_peak.py; not on the product path): the
// split-fp16 GEMM with PRE-SPLIT operands and LDS-DMA staging, used to find out what bounds the shipped kernel's K loop.
//
// Operands arrive as fp16 planes (hi, lo) in HBM -- produced by tce_split_f16_f32 here; in a planned dataflow by the
// epilogue of whichever kernel produced the tensor -- so a K slice moves HBM/L2 -> LDS with global_load_lds_dwordx4
// (no VGPR staging, no conversion VALU, no ds_write) into a 3-slot ring, a slice issued two slices before its use,
// behind counted s_waitcnt vmcnt(N) and ONE raw s_barrier per slice placed mid-slice, every fragment read prefetched
// under the previous half's MFMAs.  The LDS image is lane-linear per DMA instruction (16 rows x 64 B), so the
// bank-conflict-free XOR swizzle is applied on the SOURCE address.  256x128 tile, 8 waves (4x2) of 64x64, BK = 32,
// 48 KiB per slot.  A may also be stored K-slice-major [K/32][M][32] (a_slice) so every DMA piece is contiguous.
//
// Measured (MI355X): bit-identical results; 0-15 % faster than the shipped register-staged kernel whatever the loop
// structure.  Removing ALL DMA from the loop leaves 85 % of the time, removing the MFMAs 80 %; the bare MFMA loop of
// tools/mfma_peak.py (same 8 ds_read_b128 per 12 MFMAs, two waves per SIMD, no memory) sustains 1.6-1.7 PFLOP/s
// chip-wide (clock-limited: 2.0 PFLOP/s-equivalent per CU when only 190 CUs issue), 1.3-1.6 with a barrier per 24
// MFMAs -- so both kernels already run at 50-55 % of what the matrix pipe can sustain under this mix, and the
// remaining gap is barrier phases, not the memory path.
#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/tce_rvos.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
constexpr int BM = 256, BN = 128, BK = 32;
constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;
constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;  // 49152
constexpr int NSTAGE = 3;
constexpr int DMA_PER_WAVE = (STAGE / 1024) / 8;  // 6

__device__ __forceinline__ int swz(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// Diagnostic (tce_debug_h2_set_stamp_buffer): per wave of the first 256 workgroups, summed s_memtime ticks spent in
// [0] the counted vmcnt wait, [1] the barrier, [2] DMA issue, [3] fragment reads + MFMAs, [4] whole K loop.
__device__ long long* g_h2_stamps = nullptr;

struct H2Args {
  const _Float16 *Ah, *Al, *Wh, *Wl;
  const float *bias, *res;
  float* C;
  int M, N, K, lda, ldw, ldc, ldres, act, res_mode;
  long long a_slice;  // elements between consecutive 32-wide K slices of A (row-major: 32; slice-major [K/32][M][32]: M*32)
};

__global__ void __launch_bounds__(512, 1) gemm_h2_kernel(const H2Args p, const int tiles_m, const int tiles_n) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];  // the ONLY LDS object: base offset 0
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int nk = p.K / BK;

  // this wave's 6 DMA instructions per stage: instruction id -> (plane, first row); lane -> (row, 16-byte slot)
  const _Float16* src[DMA_PER_WAVE];
  unsigned dst[DMA_PER_WAVE];
#pragma unroll
  for (int q = 0; q < DMA_PER_WAVE; ++q) {
    const int id = wave + 8 * q;  // 0..47
    const _Float16* base;
    int row0, ld, grow0, lim, off;
    if (id < 16) { base = p.Ah; row0 = id * 16; ld = p.lda; grow0 = tm * BM; lim = p.M; off = 0; }
    else if (id < 32) { base = p.Al; row0 = (id - 16) * 16; ld = p.lda; grow0 = tm * BM; lim = p.M; off = PLANE_A; }
    else if (id < 40) { base = p.Wh; row0 = (id - 32) * 16; ld = p.ldw; grow0 = tn * BN; lim = p.N; off = 2 * PLANE_A; }
    else { base = p.Wl; row0 = (id - 40) * 16; ld = p.ldw; grow0 = tn * BN; lim = p.N; off = 2 * PLANE_A + PLANE_B; }
    const int row = row0 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);          // inverse swizzle on the source
    const int grow = min(grow0 + row, lim - 1);            // rows past the edge only feed outputs never stored
    src[q] = base + (long long)grow * ld + c * 8;
    dst[q] = (unsigned)(off + row0 * 64);
  }
  auto issue_one = [&](int kt, int slot, int q) {
    const bool is_a = (wave + 8 * q) < 32;
    glds16(src[q] + (long long)kt * (is_a ? p.a_slice : (long long)BK),
           __builtin_amdgcn_readfirstlane(dst[q] + (unsigned)(slot * STAGE)));
  };
  auto issue = [&](int kt, int slot) {
#pragma unroll
    for (int q = 0; q < DMA_PER_WAVE; ++q) {
      const bool is_a = (wave + 8 * q) < 32;
      glds16(src[q] + (long long)kt * (is_a ? p.a_slice : (long long)BK),
             __builtin_amdgcn_readfirstlane(dst[q] + (unsigned)(slot * STAGE)));
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int l31 = lane & 31, lhi = lane >> 5;

  // Fragment registers are double-buffered across K slices: the ds_reads of slice kt+1 (first k16 half) are issued
  // at the end of slice kt, so after the barrier the matrix pipe restarts from registers instead of waiting an LDS
  // round trip; the second half's reads are issued at the top and land under the first half's MFMAs.
  h16x8 fa[2][2][2], fb[2][2][2];  // [k16 half][tile][hi/lo]
  auto read_half = [&](const unsigned char* st, int ks) {
    const int c = ks * 2 + lhi;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = swz(wm * 64 + i * 32 + l31, c);
      fa[ks][i][0] = *reinterpret_cast<const h16x8*>(st + off);
      fa[ks][i][1] = *reinterpret_cast<const h16x8*>(st + PLANE_A + off);
      const int offb = swz(wn * 64 + i * 32 + l31, c);
      fb[ks][i][0] = *reinterpret_cast<const h16x8*>(st + 2 * PLANE_A + offb);
      fb[ks][i][1] = *reinterpret_cast<const h16x8*>(st + 2 * PLANE_A + PLANE_B + offb);
    }
  };
  auto mfma_half = [&](int ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[ks][j][1], fa[ks][i][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[ks][j][0], fa[ks][i][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[ks][j][0], fa[ks][i][0], acc[i][j], 0, 0, 0);
      }
  };
  // Ring discipline (one barrier per slice, placed MID-slice so that nothing the matrix pipe needs is ever fetched
  // right after it):
  //   first-half MFMAs of slice kt            (fragments already in registers)
  //   s_waitcnt vmcnt(6) ; s_barrier          slice kt+1 has landed for every wave (slice kt+2 may still fly);
  //                                           every wave also holds BOTH halves of slice kt in registers, so
  //                                           slot kt%3 is free
  //   ds_read first half of slice kt+1        lands under the second-half MFMAs
  //   second-half MFMAs of slice kt, the 6 DMA pieces of slice kt+3 (-> slot kt%3) interleaved between them
  //   ds_read second half of slice kt+1       lands under the first-half MFMAs of slice kt+1
  // A slice is issued two full slices before it is needed (96 KiB per CU in flight).
  long long* const stamps = (g_h2_stamps && blockIdx.x < 256) ? g_h2_stamps + (blockIdx.x * 8 + wave) * 8 : nullptr;
  long long t_wait = 0, t_bar = 0, t_issue = 0, t_comp = 0, t_begin = 0;
  if (stamps) t_begin = (long long)__builtin_amdgcn_s_memtime();
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  if (nk > 2) issue(2, 2);
  if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_WAVE) : "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_half(smem, 0);
  read_half(smem, 1);
  for (int kt = 0; kt < nk; ++kt) {
    long long t0 = 0, t1 = 0, t2 = 0;
    mfma_half(0);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) {
      if (stamps) t0 = (long long)__builtin_amdgcn_s_memtime();
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (stamps) t1 = (long long)__builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (stamps) t2 = (long long)__builtin_amdgcn_s_memtime();
      read_half(smem + ((kt + 1) % NSTAGE) * STAGE, 0);
    }
    // second half: 12 MFMAs with the DMA pieces of slice kt+3 dropped between them
    const bool more = kt + 3 < nk;
    const int nslot = kt % NSTAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[1][j][1], fa[1][i][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[1][j][0], fa[1][i][1], acc[i][j], 0, 0, 0);
        if (more) {
          __builtin_amdgcn_sched_barrier(0);
          issue_one(kt + 3, nslot, 2 * i + j);
          __builtin_amdgcn_sched_barrier(0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[1][j][0], fa[1][i][0], acc[i][j], 0, 0, 0);
      }
    if (more) {
      __builtin_amdgcn_sched_barrier(0);
      issue_one(kt + 3, nslot, 4);
      issue_one(kt + 3, nslot, 5);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) read_half(smem + ((kt + 1) % NSTAGE) * STAGE, 1);
    if (stamps) {
      t_wait += t1 - t0; t_bar += t2 - t1;
    }
  }
  if (stamps && lane == 0) {
    stamps[0] = t_wait; stamps[1] = t_bar; stamps[2] = t_issue; stamps[3] = t_comp;
    stamps[4] = (long long)__builtin_amdgcn_s_memtime() - t_begin;
  }
  __syncthreads();  // every wave is done reading the ring before it becomes epilogue scratch
  const bool vec_ok = tce_epi_vec_ok(p.C, p.ldc, p.res, p.ldres, p.bias, p.res_mode);
  float* wbuf = reinterpret_cast<float*>(smem) + wave * TCE_EPI_LDS_FLOATS;
#define EPI_BODY(ACT, RES)                                                                                     \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                              \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
        tce_epi_store_lds<ACT, RES>(acc[i][j], wbuf, p.bias, p.res, p.C, tm * BM + wm * 64 + i * 32,           \
                                    tn * BN + wn * 64 + j * 32, p.M, p.N, p.ldc, p.ldres, vec_ok, lane);       \
  }
  TCE_EPI_DISPATCH(p.act, p.res_mode, EPI_BODY)
#undef EPI_BODY
}

typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
struct h4pair {
  fp16x2_t a, b;
};

// x (+ add) -> fp16 planes hi = f16_rtz(x), lo = f16_rtz(x - hi); `add` has period add_rows rows (shared position map)
__global__ void __launch_bounds__(256) split_f16_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                        _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                        long long n4, int C4, long long add_rows) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
  if (add) {
    const long long row = i / C4;
    v += reinterpret_cast<const f32x4*>(add)[(row % add_rows) * C4 + (i - row * C4)];
  }
  h4pair h, l;
  h.a = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
  h.b = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
  l.a = __builtin_amdgcn_cvt_pkrtz(v[0] - (float)h.a[0], v[1] - (float)h.a[1]);
  l.b = __builtin_amdgcn_cvt_pkrtz(v[2] - (float)h.b[0], v[3] - (float)h.b[1]);
  reinterpret_cast<h4pair*>(hi)[i] = h;
  reinterpret_cast<h4pair*>(lo)[i] = l;
}

}  // namespace

// Diagnostic (tools/mfma_peak.py): what the matrix pipe sustains chip-wide under this GEMM's instruction mix, with no
// memory system involved -- mode 0: operands in registers; 1: 8 ds_read_b128 per 12 MFMAs (the fragment feed of the
// 64x64 wave tile); 2: same with three consecutive MFMAs per accumulator; 3: mode 2 + one s_barrier per 24 MFMAs.
__global__ void __launch_bounds__(512, 1) mfma_peak_kernel(float* out, int iters, int with_lds) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[65536];
  const int lane = threadIdx.x & 63;
  h16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(1.0f - i * 0.01f); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(sm)[i] = i * 1e-6f;
  __syncthreads();
  const unsigned char* base = sm + (threadIdx.x >> 6) * 4096 + lane * 16;
  for (int it = 0; it < iters; ++it) {
    if (with_lds) {  // 8 ds_read_b128 per 12 MFMAs, like the GEMM's fragment feed
      h16x8 f[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] = *reinterpret_cast<const h16x8*>(base + ((q * 1024 + it * 16) & 3071));
      if (with_lds == 1) {
#pragma unroll
        for (int q = 0; q < 12; ++q)
          acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[q & 7], f[(q + 3) & 7], acc[q & 3], 0, 0, 0);
      } else {  // the GEMM's order: three consecutive MFMAs on one accumulator
#pragma unroll
        for (int q = 0; q < 12; ++q)
          acc[q / 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[q & 7], f[(q + 3) & 7], acc[q / 3], 0, 0, 0);
      }
      if (with_lds == 3 && (it & 1)) __builtin_amdgcn_s_barrier();
    } else {
#pragma unroll
      for (int q = 0; q < 12; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q & 3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int tce_debug_mfma_peak(float* out, int32_t blocks, int32_t threads, int32_t iters, int32_t mode,
                                   tceStream stream) {
  TCE_CHECK_ARG(out && blocks > 0 && blocks <= 4096 && (threads == 256 || threads == 512) && iters > 0 && mode >= 0 && mode <= 3,
                "tce_debug_mfma_peak: bad arguments");
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters, mode);
  TCE_CHECK_LAUNCH("tce_debug_mfma_peak");
  return TCE_OK;
}

extern "C" int tce_debug_h2_set_stamp_buffer(long long* dev_buf) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_h2_stamps), &dev_buf, sizeof(dev_buf));
  if (e != hipSuccess) {
    tce_set_error("tce_debug_h2_set_stamp_buffer: %s", hipGetErrorString(e));
    return TCE_ELAUNCH;
  }
  return TCE_OK;
}

extern "C" int tce_split_f16_f32(const float* x, const float* add, void* hi, void* lo, int64_t rows, int32_t cols,
                                 int64_t add_rows, tceStream stream) {
  TCE_CHECK_ARG(x && hi && lo && rows > 0 && cols > 0 && cols % 4 == 0, "tce_split_f16_f32: bad arguments");
  const long long n4 = (long long)rows * (cols / 4);
  hipLaunchKernelGGL(split_f16_kernel, dim3(tce_cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, x, add,
                     (_Float16*)hi, (_Float16*)lo, n4, cols / 4, (long long)(add_rows > 0 ? add_rows : rows));
  TCE_CHECK_LAUNCH("tce_split_f16_f32");
  return TCE_OK;
}

extern "C" int tce_gemm_h2_f32(const void* Ah, const void* Al, const void* Wh, const void* Wl, const float* bias,
                               const float* res, float* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw,
                               int32_t ldc, int32_t ldres, int32_t act, int32_t res_mode, int64_t a_slice,
                               tceStream stream) {
  TCE_CHECK_ARG(Ah && Al && Wh && Wl && C, "tce_gemm_h2_f32: null pointer");
  TCE_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % BK == 0, "tce_gemm_h2_f32: K=%d must be a positive multiple of %d", K, BK);
  TCE_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && tce_aligned16(Ah) && tce_aligned16(Al) && tce_aligned16(Wh) &&
                    tce_aligned16(Wl),
                "tce_gemm_h2_f32: planes must be 16-byte aligned with pitches multiple of 8 halfs");
  H2Args a;
  a.Ah = (const _Float16*)Ah; a.Al = (const _Float16*)Al; a.Wh = (const _Float16*)Wh; a.Wl = (const _Float16*)Wl;
  a.bias = bias; a.res = res; a.C = C;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldres = ldres; a.act = act; a.res_mode = res_mode;
  a.a_slice = a_slice > 0 ? a_slice : BK;
  const int tiles_m = tce_cdiv(M, BM), tiles_n = tce_cdiv(N, BN);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 0);
    attr = true;
  }
  hipLaunchKernelGGL(gemm_h2_kernel, dim3(tiles_m * tiles_n), dim3(512), 0, (hipStream_t)stream, a, tiles_m, tiles_n);
  TCE_CHECK_LAUNCH("tce_gemm_h2_f32");
  return TCE_OK;
}
