/*
 * rr_oracle.h — CPU oracle for the RoaringRegex hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm (pattern -> eps-free NFA table
 * T[char][state] -> per-byte state-set stepping -> whole-string acceptance).  It exists to CHECK the
 * HIP engine; it is never linked into, imported by, or called from the product path
 * (roaringregex_amd/, include/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.
 *
 * Parity pin status (see DESIGN.md "Oracle"):
 *   - <=256 states (reference classes BitSet<1>/<2>/<4>): pinned by the known answers the survey recorded
 *     from the reference's own compiled code (SURVEY.md 8(c) table -> tests/golden/kat.json), by the table
 *     statistics it recorded (states_n / reachable / useful / byte classes), and the set
 *     primitives are pinned against the reference's own BitSet.cc built into oracle/_ref/.
 *   - >256 states (reference class roaring::Roaring): PARITY UNPINNED.  The reference is memory-unsafe and
 *     semantically wrong there (uint8 state aliasing, NFA.cc:10-11) and CRoaring is not vendored; the
 *     oracle implements the intended semantics (full-width state index) of the same algorithm.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference/src).
 */
#ifndef RR_ORACLE_H
#define RR_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rro_nfa rro_nfa;

/* Parser.cpp:161-170 (RRegex::RRegex).  Returns NULL and fills err on failure.  Patterns on which the
 * reference has undefined behaviour (empty-stack pops, bytes >= 0x80, unterminated "{m") are errors. */
rro_nfa *rro_compile(const char *pattern, char *err, size_t errcap);
void rro_free(rro_nfa *n);

uint32_t rro_states_n(const rro_nfa *n);   /* Parser.cpp:163  dry-run size          */
uint32_t rro_initial(const rro_nfa *n);    /* regex.h:81                             */
/* Parser.cpp:165-168: 1,2,4 = BitSet<W>; 0 = roaring::Roaring class (>256 states). */
int rro_set_class(const rro_nfa *n);
int rro_is_final(const rro_nfa *n, uint32_t state);
/* Row T[idx(state,c,fwd)] (NFA.cc:9-12, full-width index) as ascending state numbers.
 * Returns the row's cardinality; writes at most cap entries. */
uint32_t rro_row(const rro_nfa *n, uint32_t state, unsigned c, int fwd, uint32_t *out, uint32_t cap);

/* regex.h:156-162 + NFA.cc:72-107: whole-string acceptance of s[0..len).  A byte 0x00 or >= 0x80 inside
 * the string makes the string rejected (the reference cannot express the former and is UB on the latter). */
int rro_accepts(const rro_nfa *n, const uint8_t *s, size_t len);
/* the BitSet<2> / BitSet<4> step with the reference's vector ORs (BitSet.cc:8-21) or with scalar words: same results; default:
 * vector when the CPU has AVX2.  rro_simd() -> what is in use. */
void rro_set_simd(int on);
int rro_simd(void);

/* Batch form: split bytes[0..nbytes) on '\n' (a trailing fragment without '\n' is a line too), run
 * rro_accepts per line, write 0/1 per line.  Returns the number of lines (may exceed cap; only cap are
 * written). */
size_t rro_match_lines(const rro_nfa *n, const uint8_t *bytes, size_t nbytes, uint8_t *accept, size_t cap);
/* per line: the accepted substring [start, end) with the smallest end, then the smallest start; -1/-1 if none
 * (brute force on top of rro_accepts: short lines only) */
size_t rro_search_lines(const rro_nfa *n, const uint8_t *bytes, size_t nbytes, int32_t *start, int32_t *end, size_t cap);
/* all lazy matches per line, left to right (after a match continue at its end; one byte further after an empty one):
 * count[line], and the (start, end) pairs flattened in line order (at most cap are written); returns the total */
size_t rro_search_all(const rro_nfa *n, const uint8_t *bytes, size_t nbytes, uint32_t *count, size_t nlines_cap,
                      int32_t *start, int32_t *end, size_t cap);

/* ---- set primitives of the dense classes, exposed so tests can pin them against oracle/_ref ---- */
/* BitSet.cc:8-21 / 22-35 / 36-41 / 98-115 / 167-180 / 42-56 / 57-97, W in {1,2,4}. */
void     rro_bs_or(int W, uint64_t *a, const uint64_t *b);
void     rro_bs_and(int W, uint64_t *a, const uint64_t *b);
uint32_t rro_bs_cardinality(int W, const uint64_t *a);
uint32_t rro_bs_and_cardinality(int W, const uint64_t *a, const uint64_t *b);
void     rro_bs_add(int W, uint64_t *a, uint32_t t);
int      rro_bs_contains(int W, const uint64_t *a, uint32_t t);
void     rro_bs_shl(int W, uint64_t *dst, const uint64_t *a, int32_t rotate);
void     rro_bs_complement(int W, uint64_t *a);
/* ascending enumeration of set bits (begin/++/end protocol); returns count */
uint32_t rro_bs_iterate(int W, const uint64_t *a, int32_t *out, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif
