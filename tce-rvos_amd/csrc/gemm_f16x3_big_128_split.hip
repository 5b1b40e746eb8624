// Split-fp16 GEMM, 128x128 tile (4 waves), 3 x fp16 split arithmetic (see gemm_f16x3_big_part.inc)
#define PART_NAME p128_split
#define PART_BM 128
#define PART_BN 128
#define PART_WM 2
#define PART_SINGLE false
#include "gemm_f16x3_big_part.inc"
