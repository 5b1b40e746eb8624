// chain.hip, part 2: the token-stationary linear kernel (see the note in chain.hip)
#define TCE_CHAIN_PART 2
#include "chain.hip"
