// NFA lane engines, part 1 of 4 (see kernels_nfa.inc)
#define RRX_NFA_PART 1
#include "kernels_nfa.inc"
