// frontend.hpp — host-side pattern front end of the MI355X engine.
//
// Produces the SAME automaton (same state numbering, same transition relation, same final set) as the
// reference's RRegex::RRegex (Parser.cpp:161-170 -> build_NFA, Parser.cpp:40-159; algebra NFA.cc:42-71,
// 108-185), because bit-exact parity of the language — including the reference's quirks — is the
// contract.  The representation is our own: one edge-labelled digraph (an edge carries the 128-bit set
// of characters it is taken on) instead of the reference's per-character forward/backward row tables;
// the backward table of the reference is always the transpose of the forward one, so it is implied.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rrx {

struct CharSet {                       // characters 0..127 (the reference is 7-bit: README.md:50)
    uint64_t w[2] = {0, 0};
    void add(unsigned c) { w[c >> 6] |= 1ULL << (c & 63); }
    bool has(unsigned c) const { return c < 128 && ((w[c >> 6] >> (c & 63)) & 1); }
    bool empty() const { return !(w[0] | w[1]); }
    void operator|=(const CharSet &o) { w[0] |= o.w[0]; w[1] |= o.w[1]; }
    bool operator==(const CharSet &o) const { return w[0] == o.w[0] && w[1] == o.w[1]; }
    bool operator!=(const CharSet &o) const { return !(*this == o); }
    bool operator<(const CharSet &o) const { return w[1] != o.w[1] ? w[1] < o.w[1] : w[0] < o.w[0]; }
    CharSet inverted() const { CharSet r; r.w[0] = ~w[0]; r.w[1] = ~w[1]; return r; }
    static CharSet all() { CharSet r; r.w[0] = r.w[1] = ~0ULL; return r; }
    static CharSet single(unsigned c) { CharSet r; r.add(c); return r; }
};

struct Edge { uint32_t to; CharSet on; };

// One piece of a state's out-row.  The reference resolves "eps n->k" eagerly (NFA.cc:108-121 skip<true>): n takes a COPY of
// k's row, so x{1,n} - a right fold over n nullable copies (Parser.cpp:49-79, 123-141) - makes n^2/2 edges.  Here the copy is a
// reference: a row is a DAG of immutable-by-sharing pieces, row(s) = the union over every piece reachable from head[s] of
// its `direct` edges (labels of equal targets OR-ed).  A piece that somebody else lists as a child is `shared`; its owner
// gets a new head before it takes on another child (so the snapshot stays a snapshot), while "whatever enters n also enters
// k" (skip<false>) adds k to every piece that holds n - which is exactly every row that holds n.  The expanded graph is the
// reference's edge for edge (RefAutomaton::row, expand_row; tests/test_lowering.py compares it row by row with the oracle).
struct RowPiece {
    std::vector<Edge> direct;                     // sorted by .to
    std::vector<uint32_t> children;               // pieces whose edges this one includes
    bool shared = false;
};
constexpr uint32_t kNoPiece = UINT32_MAX;

// The reference-numbered automaton (what NFA<StateSet> holds after build_NFA).
struct RefAutomaton {
    uint32_t states_n = 0;                        // Parser.cpp:163
    uint32_t initial = 0;                         // regex.h:81
    std::vector<uint8_t> is_final;                // regex.h:177
    std::vector<RowPiece> pieces;                 // forward half of regex.h:33-35, unexpanded
    std::vector<uint32_t> head;                   // head[s]: the piece row(s) starts from, or kNoPiece (no out-edge)
    // Parser.cpp:165-168: which StateSet class the reference would instantiate (1,2,4 words; 0 = Roaring)
    int set_class() const { return states_n > 256 ? 0 : states_n > 128 ? 4 : states_n > 64 ? 2 : 1; }
    // forward row T[idx(state,c,true)] (NFA.cc:9-12, full-width index) as ascending states
    std::vector<uint32_t> row(uint32_t state, unsigned c) const;
    // out[state] of the explicit graph: every edge, sorted by target
    std::vector<Edge> expand_row(uint32_t state) const;
};

struct PatternError : std::runtime_error { using std::runtime_error::runtime_error; };
// An admitted pattern whose automaton the host pipeline will not build within its work budget (RRX_ERR_UNSUPPORTED, not a
// call that runs for minutes).
struct BudgetError : std::runtime_error { using std::runtime_error::runtime_error; };

constexpr uint32_t kMaxStates = 65536;
constexpr uint64_t kFrontEndBudget = (uint64_t)1 << 26;      // row pieces + direct edges the front end may create

// Throws PatternError with the reference's messages where the reference throws (Parser.cpp:36,155) and
// with our own where the reference has undefined behaviour (stack underflow, bytes >= 0x80, "{m" cut short).
RefAutomaton build_reference_automaton(const std::string &pattern);

}  // namespace rrx
