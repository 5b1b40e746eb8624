// Split-fp16 GEMM, 256x128 tile (8 waves), single-pass fp16 arithmetic (see gemm_f16x3_big_part.inc)
#define PART_NAME p256_single
#define PART_BM 256
#define PART_BN 128
#define PART_WM 4
#define PART_SINGLE true
#include "gemm_f16x3_big_part.inc"
