// device.hpp — host-visible interface of the HIP kernels (kernels.hip).  gfx950 only.
#pragma once
#include <cstddef>
#include <cstdint>

namespace rrx {
namespace dev {

// Geometry of the batch kernel: lane g of the grid owns the lines that START in the contiguous stripe
// bytes [g*stripe, (g+1)*stripe) and follows its last line past the stripe end.  Every lane streams its
// stripe straight from HBM into registers, kRound bytes (8 x 16 B) per round.
#ifndef RRX_THREADS
#define RRX_THREADS 1024
#endif
constexpr int kThreads = RRX_THREADS;       // lanes per workgroup: one LDS copy of the tables serves them all
// Stripe = bytes per lane: a per-corpus power of two.  Measured on the final kernel: several generations of waves
// per CU beat one generation of long-lived waves (the early finishers' slots get refilled while the slowest waves
// drain), until the per-stripe work (following the straddling line, the final flush) shows: 4 KiB is best at 8 GiB,
// 2 KiB at 1 GiB.  Automatic choice: about two million lanes, between 2 KiB and 16 KiB; explicit: 1-16 KiB.
constexpr uint32_t kMinStripe = 512, kMaxStripe = 16384, kMinAutoStripe = 2048;
constexpr size_t kTargetLanes = (size_t)1 << 21;
// (r4) Up to 512 MiB: the smallest stripe (from 512 bytes) that still leaves no more than 2^18 lanes - one generation of 256 workgroups.  A lane
// steps its stripe pair after pair, a chain of dependent LDS lookups: a 2 KiB stripe takes 55 us however small the corpus, so with 2 KiB stripes
// from 64 KiB to 128 MiB a match took 60-115 us on a handful of CUs; 512-byte stripes: 24-44 us (tools/probe/small_corpus_stripes.py).
inline uint32_t pick_stripe(size_t nbytes) {
    uint32_t s = kMinStripe;
    while (s < kMinAutoStripe && nbytes / s > ((size_t)1 << 18)) s *= 2;
    while (s < kMaxStripe && nbytes / s > kTargetLanes) s *= 2;
    return s;
}
// The size-based stripe suits lines of a few dozen bytes.  Every lane walks half a line past its stripe, so long lines
// want longer stripes (a{1,300} config, 200 B per line, 1 GiB: 4 KiB stripes +14 % over 2 KiB): double the stripe while a
// line is more than 1/16 of it.  Short lines want shorter stripes: a workgroup's 1024 lanes hold 1024 * stripe / avg_line
// lines, and beyond the 131072 its LDS result window is sure to hold (16 KiB) the result words go to memory one atomic at
// a time (k<n> lines of 5.4 bytes, 8 GiB: 4 KiB stripes 3.69 TB/s, 1 KiB stripes 4.53; profiles/r02_short_line_stripes.txt).
inline uint32_t stripe_for_lines(size_t nbytes, size_t avg_line) {
    uint32_t want = pick_stripe(nbytes);
    // ... but never below 2^18 lanes: 256 workgroups of 1024 lanes, one per CU (r4: the floor was 2^17 and 1 GiB of 440-byte lines got 8 KiB
    // stripes, i.e. 128 workgroups on 256 CUs: 2.6 TB/s against 4.4 with 4 KiB stripes; tools/probe/long_line_stripes.sh)
    while (want < kMaxStripe && avg_line * 16 > want && nbytes / (2 * (size_t)want) >= ((size_t)1 << 18)) want *= 2;
    while (want > 1024 && (avg_line + 1) * 128 < want) want /= 2;      // (512-byte stripes are for explicit requests)
    return want;
}
constexpr int kRound = 128;                      // one whole cache line per lane per round
constexpr int kMaxNfaWords = 16;                 // 512 positions per lane-resident state set
constexpr uint32_t kWideColumns = 129;           // columns 0..127 = byte values, 128 = any byte >= 0x80
constexpr uint32_t kWideMaxStates = 127;         // row byte offsets must fit 16 bits
constexpr uint32_t kClassedMaxEntries = 16384;   // row byte offsets are 16-bit: 64 KiB of 4-byte entries

struct NfaMasks {                                // passed by value -> SGPRs
    uint32_t init[kMaxNfaWords], fin[kMaxNfaWords], chain[kMaxNfaWords], self[kMaxNfaWords], excm[kMaxNfaWords];
    uint32_t cgrp[kMaxNfaWords], ctgt[kMaxNfaWords];   // add-carry rules: runs and their targets
};

struct NfaDevice {                               // tables in HBM (copied to LDS by every workgroup)
    uint32_t W = 0, nbits = 0, any_exc = 0, any_carry = 0, any_self = 0;
    NfaMasks masks;
    const uint32_t *B = nullptr;                 // [256][W]
    const uint32_t *X = nullptr;                 // [nbits][W]
};

// Group-cooperative NFA (automata beyond kMaxNfaWords*32 positions): the state set is spread over G = 8, 16 or 32 neighbouring
// lanes of a wave, K = 2 ... 8 consecutive 32-bit words per lane (word w of the set: lane w / K of the group, slot w % K), up to
// 8192 positions; a wave steps 64 / G strings.  (r4: K was fixed at 2 and a string of 2049+ positions took a whole wave.)
constexpr uint32_t kGroupMaxBits = 8192;
struct GroupNfaDevice {
    uint32_t G = 0, K = 0, nbits = 0, n_exc = 0, ncls = 0;
    uint32_t exc_mode = 0;                       // 0: exception positions anywhere; 2: position 0 (live on the first byte of a line
                                                 //   only) is the only one
    uint32_t self_slots = 0;                     // bit k: slot k of some lane holds a self loop
    uint32_t b_slots = 0;                        // bit k: slot k of some lane has a B row that is not all ones for some class >= 1
                                                 //   (the other slots are stepped by the shift alone; a class-0 byte kills the line)
    uint32_t exc_slots = 0;                      // bit k: slot k of some lane holds an exception position
    const uint32_t *masks = nullptr;             // [3][G][K]: fin, self, excm
    const uint32_t *Bcls = nullptr;              // [ncls][G][K]: positions enterable on a byte class (class 0: none)
    const uint8_t *cls = nullptr;                // [256] byte -> class (0x00 and >= 0x80: class 0)
    const uint16_t *xidx = nullptr;              // [nbits] exception row of a position (0xffff: none)
    const uint32_t *X = nullptr;                 // [n_exc][G][K]
};
// the geometry for `nbits` positions: the fewest lanes whose words hold them (more strings per wave), false beyond kGroupMaxBits
inline bool group_geometry(uint32_t nbits, uint32_t *G, uint32_t *K) {
    static const uint32_t forms[][2] = {{8, 3}, {8, 4}, {16, 3}, {16, 4}, {32, 3}, {32, 4}, {32, 5}, {32, 8}};
    for (const auto &f : forms)
        if (nbits <= f[0] * f[1] * 32u) { *G = f[0]; *K = f[1]; return true; }
    return false;
}

// Wave-resident NFA (automata beyond kGroupMaxBits positions, kernels_wave.hip): ONE WAVE holds one state set, WL 32-bit
// words per lane (WL = 1 ... 32: up to 65536 positions); exception edges are CSR lists, not rows.
constexpr uint32_t kBlockMaxBits = 65536;
struct WaveNfaDevice {
    uint32_t WL = 0, nbits = 0;
    uint32_t self_words = 0, exc_words = 0;     // bit i: word index i of some lane holds a self-loop / an exception position
    const uint32_t *masks = nullptr;             // [3][64][WL]: fin, self, excm (word w of the set = lane w / WL, index w % WL)
    const uint32_t *Bbyte = nullptr;             // [257][64][WL]: positions enterable on byte value 0..255 ('\n' an ordinary byte),
                                                 //   row 256 = the line-mode '\n' row {position 0}
    const uint32_t *xoff = nullptr;              // [nbits + 1]
    const uint32_t *xtgt = nullptr;              // targets of position p: xtgt[xoff[p] .. xoff[p+1])
    uint32_t ncls = 0;                           // sparse form only: the same rows per byte CLASS (for LDS) ...
    const uint32_t *Bcls = nullptr;              //   [ncls][64 * WL], class 0 = no position
    const uint8_t *cls = nullptr;                //   [256] byte -> class ('\n' an ordinary byte)
};
uint32_t wave_words_per_lane(uint32_t words);    // the instantiated WL that holds `words` 32-bit words (0: too many)
// The SPARSE form of the same engine (kernels_wave.hip: SparseNfa): WL then counts ROWS of 2048 positions and masks / Bbyte are
// laid out by rows - [3][WL][64] and [257][WL][64]: word w of the set = row w / 64, lane w % 64.
uint32_t sparse_rows(uint32_t words);            // the instantiated row count that holds `words` 32-bit words (0: too many)
int match_stripes_sparse_nfa(const WaveNfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                             size_t nstripes, uint32_t *accept_bits, void *stream);
int match_extents_sparse_nfa(const WaveNfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim,
                             uint8_t *accept, void *stream);

// Plain DFA (extents kernel: '\n' is an ordinary byte).
struct DfaDevice {
    uint32_t nstates = 0, ncls = 0, start = 0;
    const uint8_t *cls = nullptr;                // [256]
    const uint16_t *next = nullptr;              // [nstates][ncls]
    const uint8_t *acc = nullptr;                // [nstates]
};

constexpr size_t kPlainDfaLdsBudget = 64 * 1024;   // larger plain tables are read from HBM/L2 by the extents kernel

// Line-mode DFA tables (batch kernel): the '\n' transition of every row goes to the start row and carries
// the verdict of the line that just ended.  Entry = next row byte offset (16 bits) | nl << 16 | accept << 24.
struct LineDfaDevice {
    uint32_t nrows = 0;                          // rows (row 0 = dead)
    uint32_t stride = 0;                         // entries per row
    uint32_t start_off = 0;                      // byte offset of the start row
    uint32_t wide = 0;                           // 1: columns are byte values (kWideColumns); 0: byte classes
    uint32_t rep_log2 = 0;                       // wide form: 2^rep_log2 copies of the table interleaved dword by dword
                                                 //   (entry of copy k for logical dword i at dword i*R + k); lane l
                                                 //   reads copy l % R, i.e. only LDS banks congruent to l mod R
    uint32_t in_global = 0;                      // 1: table too large for LDS, read from HBM/L2; entry = next row's
                                                 //    first-entry index (24 bits) | nl << 30 | accept << 31
    const uint32_t *table = nullptr;             // [nrows][stride]
    const uint8_t *cls = nullptr;                // [256] (classed form only)
};

// Stride-2 line-mode table: one dependent lookup per TWO bytes.  The column of a byte pair comes from the
// state-independent table P (resolved off the critical path); T2[row][column] = next row after both bytes, with
// the number of lines that ended inside the pair (0..2) and their verdicts (oldest highest).
//   P  : u16 [128][130], P[c1*130 + c2] = byte offset of the pair's column inside a T2 row (row stride 130 = 65
//        words: the LDS bank is (c1 + c2/2) mod 32, it depends on both bytes); only for corpora without bytes >= 0x80
//   T2 : u32 [nrows][stride], entry = LDS address of the next row (16 bits) | lines << 16 | verdicts << 24,
//        2^rep_log2 interleaved copies like the wide table
struct Dfa2Device {
    uint32_t nrows = 0, stride = 0, start_off = 0, rep_log2 = 0;
    const uint16_t *P = nullptr;
    const uint32_t *T2 = nullptr;
};
constexpr uint32_t kDfa2PStride = 130;                // u16 entries per P row
constexpr uint32_t kDfa2PBytes = 128 * kDfa2PStride * 2;
// items form (trim 1): code 128 = END OF ITEM - one row more, and column 128 (one of the two pad columns of every row)
constexpr uint32_t kDfa2PItemsBytes = (129 * kDfa2PStride * 2 + 15) & ~15u;
// LDS of a stride-2 workgroup: P (32.5 KiB) + one 46 KiB region shared by T2 and the result window = 78.5 KiB, i.e. two
// 1024-lane workgroups per 160-KiB CU.  Copies of T2 are only made while they leave the window its 16 KiB.
constexpr uint32_t kDfa2RegionBytes = 46 * 1024;
constexpr uint32_t kDfa2MaxTable = kDfa2RegionBytes - 4 * 1024;       // a table this large leaves a 4 KiB window
constexpr uint32_t kDfa2TableBudget = 30 * 1024;                      // T2 with its copies

// explicit items with separators (trim 1) on the stride-2 table of their own (lower_dfa2's items form; P of kDfa2PItemsBytes):
// the arguments of items_match below
int items_match2(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, size_t nitems, const void *index, void *result, uint8_t *accept, void *stream,
                 const uint64_t *resolve_off = nullptr, const uint32_t *skip_if = nullptr);
uint32_t flush_mask_for(size_t nbytes, size_t nlines);      // the stride-2 kernel's common flush period from the mean line length
int match_stripes_dfa2(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                       size_t nstripes, uint32_t *accept, void *stream, uint32_t flush_mask = 31u);
// The sampled-table engine: the stride-2 kernel on a table with an ESCAPE state writes two bits per line into
// `wide_bits` (2 x the accept bitmap, zeroed by the caller); split_two_bit takes them apart (every word of both outputs is
// written) and counts the escaped lines; recheck_escaped_nfa lets the exact NFA lane engine decide those and ORs its accepts in.
int match_stripes_dfa2_two_bit(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                               size_t nstripes, uint32_t *wide_bits, void *stream);
// (`list`: the numbers of the escaped lines, `cap` entries; *escaped_total counts them all.  Up to `cap` escaped lines are decided
// a lane per line, more by a walk over the stripes: both kernels are queued, each reads the count and the one that is not
// needed ends at once.)
int split_two_bit(const uint32_t *wide, size_t nlines, uint32_t *accept_bits, uint32_t *escaped_bits, unsigned long long *escaped_total, uint64_t *list,
                  size_t cap, void *stream);
int recheck_escaped_nfa(const NfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base, size_t nstripes,
                        const uint32_t *escaped_bits, size_t nlines, const uint64_t *list, const unsigned long long *escaped_total, size_t cap,
                        uint32_t *accept_bits, void *stream);
// the same, stripes handed out in units of 64 inside the workgroup (units_per_wg of them per workgroup of 16 waves)
int match_units_dfa2(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                     size_t nstripes, uint32_t *accept, uint32_t units_per_wg, void *stream);

// One-pass mode (no line index yet): the stride-2 kernel writes counts[g] ('\n' per stripe, with flags) and every lane's
// verdict stream into `slabs` (onepass_slab_words() words); after scan_counts, compact_streams moves the streams to their
// place in the (zeroed) accept bitmap; words beyond cap_words are dropped (the caller learns from the line count that
// its bitmap was too small).
size_t onepass_slab_words(size_t nstripes, uint32_t stripe);
int match_onepass_dfa2(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, size_t nstripes, uint32_t *counts,
                       uint32_t *slabs, void *stream);
int match_onepass_dfa(const LineDfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, size_t nstripes, uint32_t *counts,
                      uint32_t *slabs, void *stream);
int match_onepass_nfa(const NfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, size_t nstripes, uint32_t *counts,
                      uint32_t *slabs, void *stream);
int compact_streams(const uint32_t *counts, const uint64_t *stripe_base, size_t nstripes, uint32_t stripe, const uint32_t *slabs,
                    uint32_t *accept_bits, size_t cap_words, void *stream);
// mail[0] = *total without the flag bit, mail[1] = flags ? *flags : 0, mail[2] = last_byte ? *last_byte : '\n': the few words a
// synchronous entry hands back, written into pinned device-mapped host memory by a one-lane kernel at the end of the call
int mail_results(const uint64_t *total, const uint32_t *flags, const uint8_t *last_byte, uint64_t *mail, void *stream);

// All launchers are asynchronous on `stream` and return a hipError_t value (0 = success).
int count_newlines_per_stripe(const uint8_t *bytes, size_t nbytes, uint32_t stripe, uint32_t *counts, size_t nstripes, uint32_t *flags, void *stream);
int scan_counts(const uint32_t *counts, uint64_t *base, uint64_t *chunk_sums, size_t n, void *stream);   // base[n] = total
size_t scan_scratch_words(size_t n);                                                                       // u64 words of chunk_sums
int expand_bits(const uint32_t *bits, size_t nlines, uint8_t *out, void *stream);

// accept_bits: bitmap, bit i = line i accepted; must be zeroed before the launch (the launchers do not)
int match_stripes_nfa(const NfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                      size_t nstripes, uint32_t *accept_bits, void *stream);
int match_stripes_dfa(const LineDfaDevice &p, bool clamp_high, const uint8_t *bytes, size_t nbytes, uint32_t stripe,
                      const uint64_t *stripe_base, size_t nstripes, uint32_t *accept_bits, void *stream);

int match_stripes_group_nfa(const GroupNfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                            size_t nstripes, uint32_t *accept_bits, void *stream);
int match_extents_group_nfa(const GroupNfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim,
                            uint8_t *accept, void *stream);
int match_stripes_wave_nfa(const WaveNfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                           size_t nstripes, uint32_t *accept_bits, void *stream);
int match_extents_wave_nfa(const WaveNfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim,
                           uint8_t *accept, void *stream);

// Search (the reference has acceptance only).  Per line: the match [s, e) with the smallest e, then the smallest s.
// Host tables: fwd = "any bytes, then the pattern" (never dies; accepting where some match ends), rev = the pattern read right to
// left (accepting, walking back from a match end, where a match starts).
// Stripe-wise search (kernels_search.hip).  The forward table is the LINE MODE product table (lower_search_line: one row per
// reachable pair of states plus the SKIP row - the line's first match has been found: wait for '\n' -, one column per byte
// class plus the '\n' column) in its STRIDE-2 form (lower_search_line2): one dependent lookup consumes two bytes and yields
// the four event bits of the pair.  Two layouts:
//   LDS form    (in_global = 0): P8[128][kSearchP8Stride] = 2 * pair column (a byte), T2 / T2_all[nrows][row_bytes / 2] with
//               16-bit entries (base_row + next row) << 4 | events; the kernel puts P8 at LDS address 0 and the table at
//               base_row * row_bytes, so that an entry's row field times row_bytes IS the row's LDS address;
//   global form (in_global = 1): the table stays in HBM/L2 (any size up to 256 MiB), P16[128][kSearchP16Stride] = 4 * pair
//               column in LDS, G2 / G2_all[nrows][ncols2] with 32-bit entries byte offset of the next row | events << 28.
// events = (flags of the first byte) << 2 | flags of the second; flags: 1 '\n', 2 hit, 3 hit whose match starts at the restart point.
constexpr uint32_t kSearchP8Stride = 132, kSearchP8Bytes = 128 * kSearchP8Stride;          // (33 dwords per row: odd)
constexpr uint32_t kSearchP16Stride = 130, kSearchP16Bytes = 128 * kSearchP16Stride * 2;
struct SearchChunkDevice {
    uint32_t nrows = 0, ncols2 = 0, row_bytes = 0, base_row = 0, in_global = 0;
    uint32_t start_row = 0, skip_row = 0;        // row indices (without base_row)
    uint32_t nr = 0, ncls = 0, start_r = 0;
    const uint8_t *P8 = nullptr;
    const uint16_t *P16 = nullptr;
    const uint16_t *T2 = nullptr, *T2_all = nullptr;
    const uint32_t *G2 = nullptr, *G2_all = nullptr;
    const uint16_t *rev = nullptr;               // [nr][ncls] reverse table, bit 15 = leads to an accepting state
    const uint8_t *cls = nullptr;                // [256] byte -> class
};
constexpr size_t kSearchChunkLdsBudget = 160 * 1024;     // one workgroup of 16 waves per CU: the whole LDS
size_t search_chunk_bytes();                             // bytes of text per wave (the granularity of its newline index)
size_t search_chunks_lds_bytes(const SearchChunkDevice &p);
// chunk_base: per-chunk newline prefix in the stripe_base format (bit 63: the chunk begins at the start of a line)
// clean: the text may hold bytes >= 0x80 (they cannot index the pair table: stepped as 0x00, which no pattern takes either)
// nlines: lines of the corpus (picks the build for chunks that hold more lines than the staging array: 5-byte lines)
int search_chunks(const SearchChunkDevice &p, bool clean, const uint8_t *bytes, size_t nbytes, const uint64_t *chunk_base, size_t nchunks, size_t nlines,
                  uint32_t *match_start, uint32_t *match_end, void *stream);
// all matches: count[line], then (with the caller's exclusive prefix `first`) the matches of line i at first[i], first[i] + 1, ...
int search_chunks_count(const SearchChunkDevice &p, bool clean, const uint8_t *bytes, size_t nbytes, const uint64_t *chunk_base, size_t nchunks,
                        uint32_t *count, void *stream);
int search_chunks_fill(const SearchChunkDevice &p, bool clean, const uint8_t *bytes, size_t nbytes, const uint64_t *chunk_base, size_t nchunks,
                       const uint64_t *first, uint32_t *match_start, uint32_t *match_end, void *stream);
// count and fill in one launch: first[nlines + 1] (CSR offsets of the lines' matches) and the matches themselves, slots
// >= cap counted but not written; scratch = search_all_scratch_bytes(nchunks), zeroed before the launch, holds the
// per-chunk status words, then the total (u64), then {ticket, error flag} (u32 each)
size_t search_all_scratch_bytes(size_t nchunks);
int search_chunks_all(const SearchChunkDevice &p, bool clean, const uint8_t *bytes, size_t nbytes, const uint64_t *chunk_base, size_t nchunks, size_t nlines,
                      uint64_t *first, uint32_t *match_start, uint32_t *match_end, size_t cap, void *scratch, void *stream);
// line_off[i] = offset of the first byte of line i (built once per corpus from the stripe index; only patterns that accept
// the empty string need it); nlines + 1 entries are the caller's to size, entry nlines is written only when the corpus ends in '\n'.
int build_line_offsets(const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base, size_t nstripes,
                       uint64_t *line_off, void *stream);
// all matches of a pattern that accepts "": [k, k) for k = 0 .. length of the line.  first == nullptr: count[i]; otherwise the
// slots first[i], first[i] + 1, ... are filled (slots >= cap are not written)
int empty_matches(const uint64_t *line_off, size_t nlines, uint32_t *count, const uint64_t *first, uint32_t *match_start, uint32_t *match_end,
                  void *stream, size_t cap = ~(size_t)0);

// One long string (regex.h:156-159 consumes it byte by byte): the string is cut into chunks, every chunk is stepped
// from EVERY table state at once (lane = (chunk, start state); the lanes of a chunk read the same text), which yields
// one state -> state map per chunk; maps are then composed in groups until one is left.  `scratch` holds the maps.
constexpr uint32_t kLongMaxStates = 254;         // row offsets state * 129 stay 16-bit
constexpr uint32_t kLongGroup = 128;             // maps composed per workgroup and level
// explicit items (an offsets array over one buffer) stripe-wise: see kernels_table.hip.  The table is the plain one in the
// wide line-table format with kItemColumns columns: byte values 0..127, 128 = any byte >= 0x80, kItemEndColumn = end of item
constexpr uint32_t kItemColumns = 131, kItemEndColumn = 129;      // (130 in use, 131 keeps the row stride odd)
size_t items_index_bytes(size_t nbytes, size_t nitems);  // item-end bitmap, flag, stripe base
size_t items_result_bytes(size_t nitems);                // result bitmap of one match
// resolve_base / resolve_off (one-call form, nothing known on the host): nbytes is an UPPER BOUND the index is laid out for;
// the kernels take the batch's start and length from the offsets; *flag != 0 afterwards = batch unfit for the stripe-wise
// kernel (items_match then does nothing when handed the flag as skip_if)
int items_index_build(size_t nbytes, const uint64_t *off, size_t nitems, uint32_t trim, void *index, uint32_t **flag, void *stream,
                      const uint8_t *resolve_base = nullptr, size_t min_bytes = 0);
int items_match(const LineDfaDevice &p, const uint8_t *bytes, size_t nbytes, size_t nitems, uint32_t trim, const void *index, void *result,
                uint8_t *accept, void *stream, const uint64_t *resolve_off = nullptr, const uint32_t *skip_if = nullptr);
size_t long_scratch_bytes(uint32_t nstates, size_t nbytes, uint32_t *chunk);
int match_long_dfa(const DfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t chunk, void *scratch, uint8_t *accept,
                   void *stream);
// One long string on the NFA lane engines: per chunk the rows "positions reached from position p" (lane = (chunk, p)), then
// the relations applied in order to {initial}.  `scratch` holds the rows (long_nfa_scratch_bytes).
size_t long_nfa_scratch_bytes(const NfaDevice &p, size_t nbytes, uint32_t *chunk, uint32_t *nchunks);
int match_long_nfa(const NfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t chunk, uint32_t nchunks, void *scratch, uint8_t *accept,
                   void *stream);
int match_extents_nfa(const NfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim,
                      uint8_t *accept, void *stream);
// only_if != nullptr: the kernel does nothing unless *only_if != 0 (the fallback queued behind the stripe-wise items kernel)
int match_extents_dfa(const DfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim,
                      uint8_t *accept, void *stream, const uint32_t *only_if = nullptr);

}  // namespace dev
}  // namespace rrx
