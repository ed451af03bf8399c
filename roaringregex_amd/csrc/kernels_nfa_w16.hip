// NFA lane engines, part 3 of 4 (see kernels_nfa.inc)
#define RRX_NFA_PART 3
#include "kernels_nfa.inc"
