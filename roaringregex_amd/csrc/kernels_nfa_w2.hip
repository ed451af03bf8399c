// NFA lane engines, part 0 of 4 (see kernels_nfa.inc)
#define RRX_NFA_PART 0
#include "kernels_nfa.inc"
