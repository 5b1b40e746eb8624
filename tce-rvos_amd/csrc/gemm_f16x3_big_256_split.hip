// Split-fp16 GEMM, 256x128 tile (8 waves), 3 x fp16 split arithmetic (see gemm_f16x3_big_part.inc)
#define PART_NAME p256_split
#define PART_BM 256
#define PART_BN 128
#define PART_WM 4
#define PART_SINGLE false
#include "gemm_f16x3_big_part.inc"
