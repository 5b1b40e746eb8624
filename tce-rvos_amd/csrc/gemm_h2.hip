// PROTOTYPE (tools/gemm_h2_bench.py, tools/gemm_h2_stamps.py, tools/mfma_peak.py; not on the product path): the
// split-fp16 GEMM with PRE-SPLIT operands and LDS-DMA staging, used to find out what bounds the shipped kernel's K loop.
//
// Operands arrive as fp16 planes (hi, lo) in HBM -- produced by tce_split_f16_f32 here; in a planned dataflow by the
// epilogue of whichever kernel produced the tensor -- so a K slice moves HBM/L2 -> LDS with global_load_lds_dwordx4
// (no VGPR staging, no conversion VALU, no ds_write) into a 3-slot ring, a slice issued two slices before its use,
// behind counted s_waitcnt vmcnt(N) and ONE raw s_barrier per slice.  The LDS image is lane-linear per DMA instruction (16 rows x 64 B), so the
// bank-conflict-free XOR swizzle is applied on the SOURCE address.  256x128 tile, 8 waves (4x2) of 64x64, BK = 32,
// 48 KiB per slot.  A may also be stored K-slice-major [K/32][M][32] (a_slice) so every DMA piece is contiguous.
//
// Measured (MI355X): bit-identical results; 5-15 % faster than the shipped register-staged kernel whatever the loop
// structure (barrier at the top or mid-slice, fragments prefetched or not, DMA issue interleaved or staggered between
// the two waves of a SIMD as below).  Ablation of the K loop at 24100x256xK (per 32-deep slice, K 1024 -> 2048):
//   full 1.36 us = loop skeleton 0.15 + barrier 0.03 + fragment ds_reads 0.23 + DMA 0.48 + MFMA 0.46 -- the parts ADD.
// The 16 ds_read_b128 per wave (128 KiB per CU and slice) take exactly the LDS array's 256 B/clk; the 48 KiB DMA
// fill shares that array and runs at ~100 GB/s per CU; the MFMAs (0.64 us of pipe time at 2.4 GHz) hide only partly
// and pull the clock down for everything else (tools/mfma_peak.py: 1.6-1.7 PFLOP/s sustained chip-wide).  A faster
// kernel must cut LDS bytes per MFMA (one wave per SIMD with 128x64 wave tiles: -25 % fragment reads; 256x256
// workgroup tiles: -33 % fill) and hand-interleave the three streams -- DESIGN.md section 3.1 / 8.
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
  // Ring discipline: at the top of slice kt this wave's DMAs of slice kt have landed (counted vmcnt leaves the 6 of
  // slice kt+1 in flight), the barrier publishes every wave's share and proves slot (kt+2)%3 -- slice kt-1 -- is no
  // longer read.  A wave's instruction stream is in-order, so its own DMA issue (~90 cycles per piece here) and its
  // MFMAs serialise; the overlap has to come from the OTHER wave of the SIMD.  Waves w and w+4 share a SIMD (cyclic
  // SIMD assignment), so waves 0-3 issue slice kt+2's DMA pieces BEFORE their MFMAs and waves 4-7 AFTER theirs:
  // while one of the pair feeds the memory pipe the other feeds the matrix pipe.
  long long* const stamps = (g_h2_stamps && blockIdx.x < 256) ? g_h2_stamps + (blockIdx.x * 8 + wave) * 8 : nullptr;
  long long t_wait = 0, t_bar = 0, t_issue = 0, t_comp = 0, t_begin = 0;
  if (stamps) t_begin = (long long)__builtin_amdgcn_s_memtime();
  const bool early = __builtin_amdgcn_readfirstlane(wave < 4);
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* st = smem + (kt % NSTAGE) * STAGE;
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (early && kt + 2 < nk) issue(kt + 2, (kt + 2) % NSTAGE);
    __builtin_amdgcn_sched_barrier(0);
    read_half(st, 0);
    read_half(st, 1);
    mfma_half(0);
    mfma_half(1);
    __builtin_amdgcn_sched_barrier(0);
    if (!early && kt + 2 < nk) issue(kt + 2, (kt + 2) % NSTAGE);
    __builtin_amdgcn_sched_barrier(0);
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
