// NFA lane engines, part 2 of 4 (see kernels_nfa.inc)
#define RRX_NFA_PART 2
#include "kernels_nfa.inc"
