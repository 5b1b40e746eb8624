// chain.hip, part 1: fused FFN / folded cross-attention, 3x3 convolution, patch embedding (see the note in chain.hip)
#define TCE_CHAIN_PART 1
#include "chain.hip"
