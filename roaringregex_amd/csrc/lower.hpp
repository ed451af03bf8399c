// lower.hpp — language-preserving lowering of the reference-numbered automaton into device programs.
//
// The reference steps `next = OR over s in current of T[c][s]` (NFA.cc:86-100): one row-OR per live
// state per byte.  On a 64-wide SIMD that formulation diverges per lane, so the host rewrites the SAME
// automaton into forms whose per-byte cost does not depend on the live set:
//
//   NfaProgram  "shift-and with exceptions": states are split by in-label so that every state is entered
//               on one character set (B[c] = states enterable on c); the states are laid out along a
//               maximum path cover so that most edges are "next bit" edges.  One step is
//                   next = ( ((S << 1) & CHAIN) | (S & SELF) | (((S & CGRP) + CGRP) & CTGT) | OR_{e in S & EXC} X[e] ) & B[c]
//               The add-carry term evaluates every rule "any position of a contiguous run -> the position right
//               above the run" at once (what bounded repeats x{m,n} lower to): the carry out of a run lands on
//               its target bit, which is 0 in both addends.
//               and replaces NFA.cc:86-100 (BitSet classes) and NFA.cc:77-85 (Roaring class) alike.
//   DfaProgram  the reachable state SETS of the reference's loop interned to small integers on the host
//               (subset construction + minimisation): one table lookup per byte.  Used when it stays small.
//
// Every rewrite here (trim, NUL removal, in-label split, bisimulation quotients, subset construction) keeps
// the accepted language of the reference automaton; tests/test_lowering.py replays the programs on the CPU
// and compares with the oracle.
#pragma once
#include <atomic>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "frontend.hpp"

namespace rrx {

// Useful part (reachable from initial AND able to reach a final state) of the reference automaton.
struct Trimmed {
    uint32_t n = 0;                         // 0 => empty language
    uint32_t initial = 0;
    std::vector<uint8_t> is_final;
    std::vector<std::vector<Edge>> out;     // labels never contain NUL (it cannot occur inside a string); dominated edges
                                            //   dropped (lower.cpp: trim) - every state keeps its language
    std::vector<uint32_t> ref_id;           // state id in the reference numbering
    uint32_t ncls = 1;                      // byte classes; class 0 = bytes on which nothing moves
    uint8_t cls[256];                       // byte -> class (0x00 and >= 0x80 are class 0)
    std::vector<uint8_t> cls_rep;           // one representative byte per class (cls_rep[0] unused)
};

Trimmed trim(const RefAutomaton &a);
constexpr uint64_t kTrimPairTests = (uint64_t)1 << 23;      // edge pairs the domination proofs may compare; then only the structural rule
constexpr uint64_t kTrimBudget = (uint64_t)1 << 21;          // edges the pruned rows may hold in all (then: BudgetError): the passes
                                                             //   behind trim cost 2-3 us per edge, so this is a few seconds

constexpr uint32_t kDenseExceptionBits = 4096;

struct NfaProgram {
    uint32_t W = 0;                         // 32-bit words per state set
    uint32_t nbits = 0;                     // positions in use
    uint32_t n_exc = 0;                     // positions with an exception row
    uint32_t max_exc_row_words = 0;
    std::vector<uint32_t> init, fin, chain, self, excm;   // W words each
    std::vector<uint32_t> cgrp, ctgt;                     // W words each: carry groups and their targets
    uint32_t n_carry = 0;                                 // carry groups
    std::vector<uint32_t> X;                // nbits rows of W words: extra successors of position p (dense form: only
                                            //   up to kDenseExceptionBits positions)
    std::vector<uint32_t> xoff, xtgt;       // the same in CSR form: targets of position p = xtgt[xoff[p] .. xoff[p+1])
    std::vector<uint32_t> B;                // 256 rows of W words: positions enterable on byte c
    bool accepts_empty = false;
};

struct DfaProgram {
    uint32_t nstates = 0;                   // including the dead state (id 0)
    uint32_t ncls = 0;                      // = Trimmed::ncls
    uint32_t start = 0;
    std::vector<uint8_t> accepting;         // per state
    std::vector<uint8_t> escaped;           // sampled tables only (else empty): the ESCAPE state - "the table does not know"
    std::vector<uint16_t> next;             // [state][class]
    uint8_t cls[256];
    bool accepts_empty = false;
};

// Stride-2 form of a DfaProgram: one step consumes two bytes.  Symbols are the byte classes plus '\n' (which ends
// a line: verdict of the state, restart).  Pairs of symbols with identical behaviour on every state share a column.
struct Dfa2Program {
    uint32_t nstates = 0, ncols = 0, start = 0;
    uint32_t pair_dim = 128;                // items form: 129 - code 128 is END OF ITEM, '\n' an ordinary byte
    std::vector<uint16_t> pair_col;         // [pair_dim][pair_dim]: column of the byte pair (c1, c2)
    std::vector<uint32_t> next2;            // [nstates][ncols]: next state | lines ended (0..2) << 16 | verdicts << 24
    bool accepts_empty = false;             //   (verdicts: oldest line highest)
};
// Returns false if there are more than max_cols distinct pair columns.  A sampled table (d.escaped) yields the TWO-BIT form:
// a line end shifts two result bits in, (accepted, escaped), the counts in byte 2 of an entry are bit counts.
// items = true: the table for explicit items with a separator byte each (rrx_match_extents / rrx_match_items, trim 1): every byte
// value is an ordinary symbol of the pattern - '\n' too -, what ends a line is the code 128, which the kernel puts in the place
// of the bytes the item index marks.
bool lower_dfa2(const DfaProgram &d, uint32_t max_cols, Dfa2Program &out, bool items = false);
// Profile-guided ORDER of the stride-2 table's rows and columns in LDS.  The table's entry for (state s, pair column c) sits at
// word row_slot[s] * (ncols | 1) + col_slot[c], i.e. in LDS bank (row_slot[s] * (ncols | 1) + col_slot[c]) mod 32: the order
// costs no memory and decides which entries collide when the 32 lanes of a half-wave look up 32 different (state, column)
// pairs.  `sample` = the first `bytes_per_lane` bytes of `lanes` consecutive-by-32 stripes of the corpus (lane-major): the
// lanes of a group are stepped in lockstep, as the kernel steps them, from the dead state (a stripe begins inside somebody's
// line), and the hottest rows and columns are moved, one swap at a time, to the slots that lower the mean of the worst
// bank's distinct entries per half-wave.  Returns that mean before and after.  State 0 (dead) keeps slot 0.
struct Dfa2OrderStats { double before = 0, after = 0; uint32_t half_waves = 0, evaluations = 0; };
Dfa2OrderStats order_dfa2(const Dfa2Program &d, const uint8_t *sample, uint32_t lanes, uint32_t bytes_per_lane,
                          std::vector<uint32_t> &row_slot, std::vector<uint32_t> &col_slot);

// A piece of host work that is decided ONCE per object and may run beside the caller (host only: no device call in here).
// start() or skip() moves kIdle on, everything else only READS the atomic state - the std::thread is touched by its owner
// alone (start under `mu_`, wait()).  The job runs in the deciding caller's thread or in a thread of its own (joined by
// wait() / the destructor); the state becomes kDone when it returns.
class OnceTask {
public:
    enum State { kIdle = 0, kRunning = 1, kDone = 2, kSkipped = 3 };
    OnceTask() = default;
    OnceTask(const OnceTask &) = delete;
    OnceTask &operator=(const OnceTask &) = delete;
    ~OnceTask() { wait(); }
    State state() const { return (State)state_.load(std::memory_order_acquire); }
    bool decided() const { return state() != kIdle; }
    bool start(std::function<void()> job, bool background);       // false: decided before (by another caller)
    bool skip();                // kIdle -> kSkipped; false: decided before
    void wait();                // returns when no job of this object is running in a thread of its own any more
private:
    std::atomic<int> state_{kIdle};
    std::mutex mu_;             // guards thread_
    std::thread thread_;
};
// The order search of the stride-2 table (order_dfa2) as such a task: `d` must outlive it; `apply` runs in the searching thread.
class TableOrderSearch : public OnceTask {
public:
    using Apply = std::function<void(std::vector<uint32_t> &&row_slot, std::vector<uint32_t> &&col_slot, const Dfa2OrderStats &)>;
    bool start(const Dfa2Program &d, std::vector<uint8_t> sample, uint32_t lanes, uint32_t bytes_per_lane, bool background, Apply apply);
};

// The trimmed automaton re-expressed over "positions" (a state split by the character set it is entered on; node 0
// = the initial state before any input), shrunk by bisimulation quotients and pruning of dominated edges.
// Both device programs are lowered from this one graph.
struct Reduced {
    struct Node {
        CharSet label;                      // characters this position is entered on (empty for node 0)
        bool fin = false;
        std::vector<uint32_t> follow;       // sorted unique
    };
    std::vector<Node> nodes;                // empty => empty language
    uint32_t ncls = 1;
    uint8_t cls[256];
    std::vector<uint8_t> cls_rep;
};
Reduced reduce(const Trimmed &t);

// Returns false if the position count exceeds max_bits.  gaps: a never-entered position in front of every path head
// but the first (the line-mode lane kernel shifts without a CHAIN mask).
bool lower_nfa(const Reduced &r, uint32_t max_bits, NfaProgram &out, bool allow_carry = true, bool gaps = false);
// Returns false if subset construction exceeds max_states.
bool lower_dfa(const Reduced &r, uint32_t max_states, DfaProgram &out);
// The table of an automaton whose subset construction explodes, over the sets a TEXT SAMPLE reaches (README.md:18-21:
// the live sets met on real text are few).  The sets are interned as the sample is stepped (`pieces` pieces of `piece_bytes` bytes,
// each entered at its first line start), then every transition still open is closed breadth-first while `max_states` allows;
// what stays open leads to the ESCAPE state (absorbing until the end of the line, neither accepting nor rejecting: d.escaped).
// A line that ends outside ESCAPE has exactly the reference's verdict; a line that ends in it must be decided by an exact
// engine.  Returns false if nothing could be built (an empty automaton; a sample without a line start).
struct SampledTableStats { uint32_t sets_from_sample = 0, sets_from_closure = 0, open_transitions = 0; uint64_t sample_bytes_stepped = 0, sample_escapes = 0, sample_lines = 0; };   // (sample_escapes: LINES of the sample that left the table as built)
bool lower_dfa_sampled(const Reduced &r, const uint8_t *sample, uint32_t pieces, uint32_t piece_bytes, uint32_t max_states, DfaProgram &out,
                       SampledTableStats *stats = nullptr);

// Search (the reference has acceptance only).  fwd: the DFA of "any bytes, then the pattern" - no
// byte kills it, it is accepting exactly at the positions where some match ends.  rev: the DFA of the pattern read
// right to left - walked backwards from a match end it is accepting exactly at the positions where a match starts.
bool search_dfas(const Reduced &r, uint32_t max_states, DfaProgram &fwd, DfaProgram &rev);

// Line-mode search table (stripe-wise search kernel): rows = reachable pairs (anchored state, sticky forward state) + the
// SKIP row, columns = byte classes + '\n'.  Entry = next row | flags: kSearchNewline (the byte was '\n': next row = start),
// kSearchHit (the byte completes the line's first match: next row = SKIP), kSearchAnchored (with kSearchHit: the line's
// prefix up to here is itself accepted, i.e. the match with the smallest start begins at the line start).
constexpr uint32_t kSearchNewline = 1u << 16, kSearchHit = 1u << 17, kSearchAnchored = 1u << 18;
struct SearchLineProgram {
    uint32_t nrows = 0, ncols = 0, start = 0, skip = 0;
    std::vector<uint32_t> table;            // [nrows][ncols]
};
// anchored == nullptr: the forward table alone (rows = its states + SKIP, no hit carries kSearchAnchored: every match start is walked back to)
bool lower_search_line(const DfaProgram &fwd, const DfaProgram *anchored, uint32_t max_rows, SearchLineProgram &out);

// Stride-2 form of the line-mode search table (the forward pass of the stripe-wise kernel consumes two bytes per dependent
// lookup, as the match kernel's stride-2 table does): symbols = the table's columns (byte classes, '\n' last), one column
// per PAIR of symbols, pairs that act alike on every row - in the first-match form AND in the restart form - share one.
// Entry = next row | events << 24, events = (flags of the first byte) << 2 | flags of the second, flags as the kernel's event
// words hold them: 0 nothing, 1 the byte was '\n', 2 it completed a match, 3 a match that starts at the restart point.
// first: after a hit the row is SKIP until the '\n'; all: a hit leads back to the start row (the search goes on behind it).
struct SearchLine2Program {
    uint32_t nrows = 0, ncols = 0, start = 0, skip = 0;
    std::vector<uint16_t> pair_col;         // [128][128]: column of the byte pair (c1, c2); bytes >= 0x80 are stepped as 0x00
    std::vector<uint32_t> first, all;       // [nrows][ncols]
};
// `column`: byte -> column of `s` (256 entries; '\n' -> the last column).  Returns false beyond max_cols distinct pair columns.
bool lower_search_line2(const SearchLineProgram &s, const uint32_t *column, uint32_t max_cols, SearchLine2Program &out);

}  // namespace rrx
