// ref_bitset_shim.cpp — TEST INFRASTRUCTURE.  A driver (not a stand-in): it includes the reference's OWN
// BitSet.h and is linked with the reference's OWN BitSet.cc, both compiled from where they lie under
// /root/reference/src (see oracle/Makefile; outputs go to oracle/_ref/ only, nothing is copied).
// BitSet.cc is the only reference translation unit that builds without the un-vendored CRoaring
// (regex.h:8 includes ../../CRoaring/roaring.hh), so it is the only part of the reference that can be
// executed here; it pins the oracle's set primitives (rro_bs_*).
#include "BitSet.h"
#include <cstdint>
#include <cstring>

template <int W> static BitSet<W> load(const uint64_t *w) { BitSet<W> b; std::memcpy(b.words, w, sizeof(uint64_t) * W); return b; }
template <int W> static void store(uint64_t *w, const BitSet<W> &b) { std::memcpy(w, b.words, sizeof(uint64_t) * W); }

#define DISPATCH(W, BODY1, BODY2, BODY4) do { if ((W) == 1) { BODY1; } else if ((W) == 2) { BODY2; } else { BODY4; } } while (0)

template <int W> static void t_or(uint64_t *a, const uint64_t *b) { BitSet<W> x = load<W>(a); x |= load<W>(b); store<W>(a, x); }
template <int W> static void t_and(uint64_t *a, const uint64_t *b) { BitSet<W> x = load<W>(a); x &= load<W>(b); store<W>(a, x); }
template <int W> static uint32_t t_card(const uint64_t *a) { return load<W>(a).cardinality(); }
template <int W> static uint32_t t_andcard(const uint64_t *a, const uint64_t *b) { BitSet<W> x = load<W>(a); return (uint32_t)x.and_cardinality(load<W>(b)); }
template <int W> static void t_add(uint64_t *a, uint32_t t) { BitSet<W> x = load<W>(a); x.add(t); store<W>(a, x); }
template <int W> static int t_contains(const uint64_t *a, uint32_t t) { BitSet<W> x = load<W>(a); return x.contains(t); }
template <int W> static void t_shl(uint64_t *d, const uint64_t *a, int32_t r) { BitSet<W> x = load<W>(a); BitSet<W> y = x + r; store<W>(d, y); }
template <int W> static void t_compl(uint64_t *a) { BitSet<W> x = load<W>(a); x.complement(); store<W>(a, x); }
template <int W> static uint32_t t_iter(const uint64_t *a, int32_t *out, uint32_t cap) {
    BitSet<W> x = load<W>(a);
    uint32_t k = 0;
    for (typename BitSet<W>::const_iterator i = x.begin(); i < x.end(); ++i) { if (k < cap) out[k] = *i; k++; }
    return k;
}

extern "C" {
void ref_bs_or(int W, uint64_t *a, const uint64_t *b) { DISPATCH(W, t_or<1>(a, b), t_or<2>(a, b), t_or<4>(a, b)); }
void ref_bs_and(int W, uint64_t *a, const uint64_t *b) { DISPATCH(W, t_and<1>(a, b), t_and<2>(a, b), t_and<4>(a, b)); }
uint32_t ref_bs_cardinality(int W, const uint64_t *a) { DISPATCH(W, return t_card<1>(a), return t_card<2>(a), return t_card<4>(a)); return 0; }
uint32_t ref_bs_and_cardinality(int W, const uint64_t *a, const uint64_t *b) { DISPATCH(W, return t_andcard<1>(a, b), return t_andcard<2>(a, b), return t_andcard<4>(a, b)); return 0; }
void ref_bs_add(int W, uint64_t *a, uint32_t t) { DISPATCH(W, t_add<1>(a, t), t_add<2>(a, t), t_add<4>(a, t)); }
int ref_bs_contains(int W, const uint64_t *a, uint32_t t) { DISPATCH(W, return t_contains<1>(a, t), return t_contains<2>(a, t), return t_contains<4>(a, t)); return 0; }
void ref_bs_shl(int W, uint64_t *d, const uint64_t *a, int32_t r) { DISPATCH(W, t_shl<1>(d, a, r), t_shl<2>(d, a, r), t_shl<4>(d, a, r)); }
void ref_bs_complement(int W, uint64_t *a) { DISPATCH(W, t_compl<1>(a), t_compl<2>(a), t_compl<4>(a)); }
uint32_t ref_bs_iterate(int W, const uint64_t *a, int32_t *out, uint32_t cap) { DISPATCH(W, return t_iter<1>(a, out, cap), return t_iter<2>(a, out, cap), return t_iter<4>(a, out, cap)); return 0; }
}
