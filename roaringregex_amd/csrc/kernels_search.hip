// kernels_search.hip — stripe-wise search (rrx_search_corpus): for every line the accepted substring
// with the smallest end, then the smallest start.  Shared device code: kernels_common.hpp.
//
// Geometry: a WAVE owns a contiguous chunk of 64 * S bytes, lane l its sub-stripe [l*S, (l+1)*S) (S = 256: two cache
// lines per lane, one after the other in registers).  A line belongs to the lane its first byte lies in.  Per lane:
//   1. forward pass over its S bytes with the line-mode forward table ("any bytes, then the pattern"; an entry carries two
//      flags per byte: the byte was '\n' / the byte completed the first match of its line, after which the row is SKIP
//      until the next '\n') in its STRIDE-2 form: the pair's column comes from a table indexed by the two bytes (its read does
//      not wait for the state), then one dependent lookup consumes both bytes and yields their four event bits - 2 LDS
//      gathers per PAIR (round 3: per byte; the LDS pipe is what bounds the pass), no branch; the flags are shifted into
//      event words, 2 bits per byte.  FORM kLdsForm: the table in LDS with 16-bit entries; kGlobalForm: the table in
//      HBM/L2 (product tables beyond the LDS: any size), only the pair table in LDS;
//   2. a wave prefix sum of the lanes' '\n' counts numbers the lines (the chunk's first line number comes from a per-chunk
//      newline index, as the match kernel's stripes have);
//   3. the hit events are visited in byte order (first match: the hits only - a hit's line number and line start are a
//      popcount and a find-first over the '\n' events in front of it; all matches: every event, the '\n' carry the per-line
//      counts and slots).  A hit whose start is known (the line start / the end of the previous match: a flag of the table)
//      leaves (start, end) in the wave's LDS staging array at the line's number; any other leaves a JOB in the wave's pool;
//   4. a lane whose last line is still undecided at the end of its sub-stripe follows it into the next lanes' bytes;
//      whenever the pool would run over - and at the end - the wave walks 64 jobs at a time, a job per lane, back from the
//      match end with the reverse table (four bytes per turn, text re-read from L1/L2 as aligned words);
//   5. the wave writes its lines' results, consecutive lanes consecutive lines: whole sectors, where one lane per stripe
//      appending 8 bytes per line left them as partial writes (10 ms for the line offsets alone in the first version).
// ALL matches of every line (rrx_search_all_count / _fill) run the same kernel in two more modes with the "restart" form
// of the table (a hit leads back to the start row instead of SKIP: the search goes on right after the match): kCount
// counts the hits per line; kFill numbers a line's matches from first[line] (the caller's exclusive prefix of the counts),
// bounds every walk back by the end of the previous match, and stages the wave's matches - a contiguous range of slots.
#include "kernels_common.hpp"

namespace rrx {
namespace dev {
namespace {

constexpr int kSearchWaves = 16;                // waves per workgroup: ONE workgroup per CU (128 VGPRs: four waves per SIMD), one table copy in LDS serves them
constexpr int kSearchS = 256;                   // bytes per lane: two 128-byte rounds (the per-lane costs that do not scale with
                                                // the bytes - following the last line, numbering, write-out - are paid half as often)
constexpr uint32_t kSearchChunk = 64 * kSearchS;
constexpr uint32_t kMaxStageLines = 2048;       // staged lines per wave (what the tables leave of the LDS budget, at most this); lines beyond go to memory directly
constexpr uint32_t kNone = 0xffffffffu;
constexpr uint32_t kPool = 128;                 // walk jobs a wave collects before it walks 64 of them, a job per lane (four words each)
constexpr uint32_t kPoolWords = kPool * 4;
constexpr uint32_t kDirect = 0xfffffffeu;       // staged entry: the result did not fit 16 + 16 bits and went to memory directly

typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
typedef const __attribute__((address_space(3))) uint16_t *lds_u16_ptr;

enum { kLdsForm = 0, kGlobalForm = 1 };
struct SearchLds {
    // reverse table, as the kernel lays it out in LDS: entry = byte offset of the next state's row | 1 if that state is
    // accepting (rows are 2 * K bytes: even); row 0 = dead (every entry leads back to it); cls2 = byte -> 2 * class
    uint32_t rev_base, cls2_base;               // LDS addresses
    uint32_t start_row;                         // row offset of the reverse start state
    uint32_t start_e, skip_e;                   // forward table: the entry values "at the start row" / "at the SKIP row"
};
typedef const __attribute__((address_space(3))) uint8_t *lds_u8_ptr;

// one reverse step: r = row offset | accept bit of the state reached so far, c = the byte in front
__device__ __forceinline__ uint32_t reverse_step(const SearchLds &t, uint32_t r, uint32_t cb) {
    return *reinterpret_cast<lds_u16_ptr>((r & 0xfffeu) + cb + t.rev_base);
}
// first byte position s in [lo, e] such that bytes[s, e) is accepted (e itself if only the empty string is: cannot
// happen here, patterns that accept the empty string take another path).  The plain form: a byte per turn (a lane whose
// job queue is full and cannot wait for the wave's common walk - rare since the queue is drained whenever a lane's fills up)
__device__ __forceinline__ size_t reverse_walk(const SearchLds &t, const uint8_t *__restrict__ bytes, size_t lo, size_t e) {
    uint32_t r = t.start_row;
    size_t best = e, k = e;
    while (k > lo) {
        k--;
        const uint32_t cb = *reinterpret_cast<lds_u8_ptr>(t.cls2_base + (uint32_t)bytes[k]);
        r = reverse_step(t, r, cb);
        if (r == 0u) break;                                       // row 0 is dead (and not accepting)
        if (r & 1u) best = k;
    }
    return best;
}

enum { kFirst = 0, kCount = 1, kFill = 2, kAll = 3 };

// kAll (rrx_search_all): count and fill in ONE launch.  A wave counts the matches of its chunk (the replay of its events
// and the walk along its last line, nothing written), learns the number of matches in all chunks before its own by a
// decoupled look-back over per-chunk status words (Merrill & Garland's single-pass scan: every chunk publishes its
// count, then the sum through itself; a wave sums its predecessors' counts back to the nearest published sum), and
// replays its events a second time, now placing every match.  Chunks are handed out by a ticket counter, so every chunk
// before mine belongs to a wave that is already running: the look-back never waits for a wave that has not started.
struct SearchAllArgs {
    uint64_t *first_out;        // [nlines + 1] slot of the first match of every line (CSR offsets), written here
    uint64_t *status;           // [nchunks] zeroed before the launch
    uint32_t *ticket;           // [0] next chunk, [1] error flag (look-back gave up); zeroed before the launch
    uint64_t *total;            // number of matches
    uint64_t cap;               // slots the caller's match arrays hold: matches beyond are counted, not written
    uint64_t nlines;
};
constexpr uint64_t kStAggregate = 1ull << 62, kStInclusive = 2ull << 62, kStFlags = 3ull << 62;
constexpr uint64_t kLookbackTicks = 10ull * 100000000ull;     // 10 s of the 100 MHz wall clock: a chunk whose first line runs on
                                                              // for megabytes is counted by ONE lane, and everybody behind waits
// -> matches in the chunks before `chunk`; publishes `mine` (the chunk's own count) and then the sum through the chunk
__device__ __forceinline__ uint64_t chunk_lookback(uint64_t *status, uint32_t *ticket, size_t chunk, uint64_t mine, int lane) {
    if (lane == 0) __hip_atomic_store(&status[chunk], kStAggregate | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t excl = 0;
    int64_t pos = (int64_t)chunk;                                 // the window: chunks [pos - 64, pos), lane 0 the nearest
    uint32_t spins = 0;
    uint64_t t0 = 0;
    for (;;) {
        const int64_t idx = pos - 1 - lane;
        const uint64_t st = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kStInclusive;
        const uint64_t incl = __ballot((st & kStFlags) == kStInclusive);
        const int first_incl = incl ? __ffsll((long long)incl) - 1 : 64;
        const uint64_t need = first_incl < 63 ? ((2ull << first_incl) - 1ull) : ~0ull;      // lanes 0 .. first_incl
        if (__ballot((st & kStFlags) == 0) & need) {              // a chunk on the way has not published yet
            bool give_up = false;
            if ((++spins & 255u) == 0) {
                const uint64_t now = wall_clock64();
                if (!t0) t0 = now;
                give_up = now - t0 > kLookbackTicks || __hip_atomic_load(&ticket[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            }
            if (give_up) {                                        // never spin for ever: flag the launch as failed and drain
                if (lane == 0) atomicOr(&ticket[1], 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        uint64_t v = lane <= first_incl ? (st & ~kStFlags) : 0ull;
#pragma unroll
        for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d, 64);
        excl += v;
        if (first_incl < 64) break;
        pos -= 64;
    }
    if (lane == 0) __hip_atomic_store(&status[chunk], kStInclusive | (excl + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// MODE kFirst: match_start/match_end[line] = first match of the line (kNone: no match).
// MODE kCount: match_start[line] = number of matches of the line (match_end unused).
// MODE kFill : match_start/match_end[first[line] + k] = k-th match of the line.
// MODE kAll  : the same, with first[] produced here as well (all.first_out) and the slots bounded by all.cap.
// MULTI (kFirst only): built for corpora whose 16-KiB chunks hold more lines than the staging array (5-byte lines: 3277 against 1728):
// the lines are taken in WINDOWS of kStageLines ordinals, each parked, taken a line per lane and written out in turn.  That keeps the
// event words alive through the follow loop - 16 registers, 14-20 % of the rate on every other corpus - so it is a build of its own,
// chosen by the launcher from the corpus' mean line length; the plain build takes a chunk that overflows through the wave-wide loop.
template <int MODE, int FORM, bool MULTI = false>
__global__ __launch_bounds__(kSearchWaves * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) void search_chunks_kernel(SearchChunkDevice prog, uint32_t clean, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                          const uint64_t *__restrict__ chunk_base, size_t nchunks,
                                                                          const uint64_t *__restrict__ first,
                                                                          uint32_t *__restrict__ match_start, uint32_t *__restrict__ match_end,
                                                                          uint32_t kStageLines, SearchAllArgs all) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // ---- tables into LDS, once per workgroup: its waves then take chunk after chunk (a chunk is only 16 KiB of text, the
    // tables can be several times that).  The pair table comes first, at LDS address 0, so that the pair's index IS its
    // address (the dynamic LDS starts at address 0: no static LDS in this kernel, the launcher checks it on the host); the
    // forward table (LDS form) starts at a multiple of its row size: an entry's row field times the row size is an address.
    const uint32_t rev_words = (prog.nr * prog.ncls + 1) / 2;
    uint32_t tab_end;                                             // bytes of LDS the forward tables take
    const uint32_t *G = nullptr;                                  // global form: the table in HBM/L2
    if constexpr (FORM == kLdsForm) {
        uint32_t *P = reinterpret_cast<uint32_t *>(smem);
        for (uint32_t i = threadIdx.x; i < kSearchP8Bytes / 4; i += blockDim.x) P[i] = reinterpret_cast<const uint32_t *>(prog.P8)[i];
        const uint32_t t_base = prog.base_row * prog.row_bytes, t_words = prog.nrows * (prog.row_bytes / 4);
        uint32_t *T = reinterpret_cast<uint32_t *>(smem + t_base);
        const uint32_t *Tsrc = reinterpret_cast<const uint32_t *>(MODE == kFirst ? prog.T2 : prog.T2_all);
        for (uint32_t i = threadIdx.x; i < t_words; i += blockDim.x) T[i] = Tsrc[i];
        tab_end = t_base + 4u * t_words;
    } else {
        uint32_t *P = reinterpret_cast<uint32_t *>(smem);
        for (uint32_t i = threadIdx.x; i < kSearchP16Bytes / 4; i += blockDim.x) P[i] = reinterpret_cast<const uint32_t *>(prog.P16)[i];
        tab_end = kSearchP16Bytes;
        G = MODE == kFirst ? prog.G2 : prog.G2_all;
    }
    uint32_t *R = reinterpret_cast<uint32_t *>(smem + tab_end);
    uint32_t *C = R + rev_words;                                  // 64 words
    uint32_t *pool_all = C + 64;                                  // [wave][kPoolWords]: the waves' walk jobs
    uint32_t *stage = pool_all + kSearchWaves * kPoolWords;       // [wave][kArrays][kStageLines]
    constexpr uint32_t kArrays = (MODE == kFill || MODE == kAll) ? 2 : 1;     // results packed start | end << 16 (kCount: the count); kFill / kAll: + slot bases
    constexpr uint32_t kStageInit = MODE == kCount ? 0u : kNone;  // a line without a match: "none" / zero matches
    // the reverse table changes form on the way in: next state | accepting << 15  ->  row offset | accepting (see SearchLds);
    // the byte -> class map is doubled (classes are < 128)
    const uint32_t rev_row = 2u * prog.ncls;
    for (uint32_t i = threadIdx.x; i < rev_words; i += blockDim.x) {
        const uint32_t w = reinterpret_cast<const uint32_t *>(prog.rev)[i];
        const uint32_t a = (w & 0x7fffu) * rev_row | ((w >> 15) & 1u), b = ((w >> 16) & 0x7fffu) * rev_row | (w >> 31);
        R[i] = a | b << 16;
    }
    for (uint32_t i = threadIdx.x; i < 64; i += blockDim.x) C[i] = (reinterpret_cast<const uint32_t *>(prog.cls)[i] << 1) & 0xfefefefeu;
    for (uint32_t i = threadIdx.x; i < kSearchWaves * kArrays * kStageLines; i += blockDim.x) stage[i] = kStageInit;
    __syncthreads();
    SearchLds t;
    t.rev_base = tab_end; t.cls2_base = t.rev_base + 4u * rev_words; t.start_row = prog.start_r * rev_row;
    if constexpr (FORM == kLdsForm) { t.start_e = (prog.base_row + prog.start_row) << 4; t.skip_e = (prog.base_row + prog.skip_row) << 4; }
    else { t.start_e = prog.start_row * prog.ncols2 * 4u; t.skip_e = prog.skip_row * prog.ncols2 * 4u; }
    const uint32_t row_bytes = prog.row_bytes;
    // ---- the forward step.  pair_index: where the column of each of the two byte pairs of a text word stands in the pair table
    // (= its LDS address); pair_column: that column; step_pair: the dependent lookup, the pair's four event bits shifted into acc.
    //     LDS form    i = c1 * 132 + c2;  col = P8[i] (2 * column);   e = T2[(e >> 4) * row_bytes + col] (u16);  acc = acc << 4 | (e & 15)
    //     global form i = 2 c1 * 130 + 2 c2;  col = P16[i] (4 * column);  e = G2[(e & 0x0fffffff) + col];      acc = {acc, e} >> 28
    // (asm volatile keeps the lookups of a few pairs ahead of their steps and no more: left to itself the scheduler hoists every
    // column lookup of a round and spills their results.)  Bytes >= 0x80 cannot index the pair table: `clean` (a corpus that
    // holds any) rewrites them to 0x00 - no pattern takes either - under a wave-uniform branch per 16 bytes.
    auto clean_word = [](uint32_t w) -> uint32_t { const uint32_t hi = (w & 0x80808080u) >> 7; return w & ~(hi * 0xffu); };
    auto pair_index = [&](uint32_t w, uint32_t &ia, uint32_t &ib) {
        const uint32_t x = FORM == kGlobalForm ? w << 1 : w;
        const uint32_t stride = FORM == kGlobalForm ? kSearchP16Stride : kSearchP8Stride;
        uint32_t ta, tb;
        // (the two multiplies first: an SDWA read of a register the instruction before wrote costs a wait state)
        asm volatile("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(ta) : "v"(x), "v"(stride));
        asm volatile("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(tb) : "v"(x), "v"(stride));
        asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(ia) : "v"(ta), "v"(x));
        asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(ib) : "v"(tb), "v"(x));
    };
    auto pair_column = [&](uint32_t i) -> uint32_t {
        if constexpr (FORM == kLdsForm) return *reinterpret_cast<lds_u8_ptr>(i);
        else return *reinterpret_cast<lds_u16_ptr>(i);
    };
    auto step_pair = [&](uint32_t &e, uint32_t col, uint32_t &acc) {
        uint32_t a_;
        if constexpr (FORM == kLdsForm) {
            asm volatile("v_lshrrev_b32 %0, 4, %1\n\tv_mad_u32_u24 %0, %0, %2, %3" : "=&v"(a_) : "v"(e), "s"(row_bytes), "v"(col));
            e = *reinterpret_cast<lds_u16_ptr>(a_);
            asm volatile("v_and_b32 %0, 15, %2\n\tv_lshl_or_b32 %1, %1, 4, %0" : "=&v"(a_), "+v"(acc) : "v"(e));
        } else {
            asm volatile("v_and_b32 %0, 0x0fffffff, %1\n\tv_add_u32 %0, %0, %2" : "=&v"(a_) : "v"(e), "v"(col));
            e = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(G) + a_);
            asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(acc) : "v"(e));
        }
    };

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t *st_s = stage + (size_t)wave * kArrays * kStageLines;
    uint32_t *st_f = st_s + kStageLines;                          // kFill / kAll: first[line] - first[the wave's first line]
    uint32_t *const pool = pool_all + (size_t)wave * kPoolWords;
    auto next_chunk = [&](size_t prev, bool start) -> size_t {
        if constexpr (MODE == kAll) {                             // by ticket (see SearchAllArgs)
            uint32_t tk = 0;
            if (lane == 0) tk = atomicAdd(&all.ticket[0], 1u);
            return (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
        }
        return start ? (size_t)blockIdx.x * kSearchWaves + wave : prev + (size_t)gridDim.x * kSearchWaves;
    };
    for (size_t chunk = next_chunk(0, true); chunk < nchunks; chunk = next_chunk(chunk, false)) {
    const size_t cstart = chunk * (size_t)kSearchChunk;
    const size_t cend = cstart + kSearchChunk < nbytes ? cstart + kSearchChunk : nbytes;      // end of the chunk's data
    const uint64_t cb = chunk_base[chunk];
    const uint64_t base_line = line_of(cb);
    const bool chunk_fresh = (cb & kFreshStripe) != 0;
    const size_t my = cstart + (size_t)lane * kSearchS;
    const size_t my_end = my + kSearchS < cend ? my + kSearchS : (my < cend ? cend : my);
    const uint32_t vlen = (uint32_t)(my_end - my);                // my bytes: kSearchS except at the end of the corpus

    // ---- 1. forward pass over my bytes.  Every lane starts as if a line started at its first byte; whether one does - the byte in
    // front of it, the last of the lane before - is read off that lane's EVENTS afterwards, and a lane that began inside a line then
    // drops the hits it saw before its first '\n' (what starting in SKIP gave for nothing).  (r4) Until then every lane read its
    // last byte up front: the second cache line of its 256 bytes, touched a round before it is consumed and gone again by then -
    // the first-match kernel fetched 1.57 x the text, FETCH_SIZE, tools/probe/traffic_check.sh.)
    uint32_t e = vlen ? t.start_e : t.skip_e;
    constexpr int kEv = kSearchS / 16;
    uint32_t ev[kEv];                                             // 2 bits per byte, the first byte of a word highest
    {
        // my bytes, 128 at a time in registers.  At the end of the corpus a lane holds fewer: whole 16-byte slots inside the
        // data are loaded as such, the slot the data ends in byte by byte, the rest is zero; events behind the data are masked.
        auto round = [&](auto round_index) {
        constexpr int kR = decltype(round_index)::value;
        const size_t rb = my + (size_t)128 * kR;
        const uint32_t rlen = vlen > 128u * kR ? (vlen - 128u * kR < 128u ? vlen - 128u * kR : 128u) : 0u;
        TextRound<8> text;
        if (rlen == 128u) text.load(reinterpret_cast<const uint4 *>(bytes + rb));
        else {
            uint32_t off = 0;
            text.for_each_slot_mut([&](uint4 &v) {
                if (off + 16 <= rlen) v = *reinterpret_cast<const uint4 *>(bytes + rb + off);
                else {
                    uint32_t w[4] = {0, 0, 0, 0};
                    for (uint32_t k = off; k < rlen && k < off + 16; k++) w[(k - off) >> 2] |= (uint32_t)bytes[rb + k] << (8 * ((k - off) & 3));
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
                off += 16;
            });
        }
        int slot = 8 * kR;
        text.for_each_slot([&](const uint4 &v0) {
            uint4 v = v0;
            if (clean && __builtin_amdgcn_ballot_w64(((v.x | v.y | v.z | v.w) & 0x80808080u) != 0)) {
                v.x = clean_word(v.x); v.y = clean_word(v.y); v.z = clean_word(v.z); v.w = clean_word(v.w);
            }
            uint32_t acc = 0;
            // the columns of four pairs (two text words) are looked up ahead of their four dependent steps
            {
                uint32_t i0, i1, i2, i3;
                pair_index(v.x, i0, i1); pair_index(v.y, i2, i3);
                const uint32_t c0 = pair_column(i0), c1 = pair_column(i1), c2 = pair_column(i2), c3 = pair_column(i3);
                step_pair(e, c0, acc); step_pair(e, c1, acc); step_pair(e, c2, acc); step_pair(e, c3, acc);
            }
            {
                uint32_t i0, i1, i2, i3;
                pair_index(v.z, i0, i1); pair_index(v.w, i2, i3);
                const uint32_t c0 = pair_column(i0), c1 = pair_column(i1), c2 = pair_column(i2), c3 = pair_column(i3);
                step_pair(e, c0, acc); step_pair(e, c1, acc); step_pair(e, c2, acc); step_pair(e, c3, acc);
            }
            // (slot is a compile-time constant after inlining: the lambda is expanded once per slot)
            ev[slot++] = acc;
        });
        };
        round(std::integral_constant<int, 0>{});
        round(std::integral_constant<int, 1>{});
        if (vlen != kSearchS) {                                   // events behind the end of the data do not exist
#pragma unroll
            for (int i = 0; i < kEv; i++) {
                const int valid = (int)vlen - 16 * i;             // bytes of this word inside the data
                if (valid <= 0) ev[i] = 0;
                else if (valid < 16) ev[i] &= ~0u << (2 * (16 - valid));
            }
        }
    }
    const bool my_last_nl = vlen == kSearchS ? (ev[kEv - 1] & 3u) == 1u : (vlen ? bytes[my_end - 1] == '\n' : false);
    const uint32_t prev_last_nl = __shfl_up((uint32_t)my_last_nl, 1, 64);
    const bool fresh = vlen && (lane == 0 ? chunk_fresh : prev_last_nl != 0);
    if (vlen && !fresh) {                                         // I began inside somebody's line: no hit of mine before my first '\n'
        bool open = true;
#pragma unroll
        for (int i = 0; i < kEv; i++) {
            if (open) {
                const uint32_t nlm = ev[i] & ~(ev[i] >> 1) & 0x55555555u;      // fields equal to 1
                const uint32_t hm = ev[i] & 0xaaaaaaaau, hits = hm | (hm >> 1);
                if (nlm) {
                    const uint32_t top = 31u - (uint32_t)__clz((int)nlm);        // low bit of the first '\n' field (the first byte is highest)
                    ev[i] &= ~(hits & (top >= 30u ? 0u : ~0u << (top + 2u)));
                    open = false;
                } else ev[i] &= ~hits;
            }
        }
    }
    // ---- 2. number the lines: '\n' counts, prefix over the lanes
    uint32_t nl = 0;
#pragma unroll
    for (int i = 0; i < kEv; i++) nl += __popc(ev[i] & ~(ev[i] >> 1) & 0x55555555u);      // fields equal to 1
    uint32_t incl = nl;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    const uint32_t total_nl = __shfl(incl, 63, 64);
    uint32_t ord = incl - nl;                                     // ordinal (within the chunk) of the line my first byte is in

    // ---- 3. results.  kFirst / kCount: staged by line ordinal.  kFill: the wave's matches are the slots first[its first
    // line] ... in line order; the slot base of every staged line comes from `first`, relative to the wave's first slot.
    const bool ends_on_nl = cend - cstart == kSearchChunk ? __shfl((uint32_t)my_last_nl, 63, 64) != 0 : bytes[cend - 1] == '\n';
    const uint32_t lo_ord = chunk_fresh ? 0u : 1u;                // ordinals lo..hi start in this chunk: they are the wave's
    const int64_t hi_ord = ends_on_nl ? (int64_t)total_nl - 1 : (int64_t)total_nl;
    uint64_t F0 = 0;                                              // kFill: first slot of the wave
    if constexpr (MODE == kFill) {
        if ((int64_t)lo_ord <= hi_ord) {
            F0 = first[base_line + lo_ord];
            for (int64_t j = lo_ord + lane; j <= hi_ord && j < (int64_t)kStageLines; j += 64) st_f[j] = (uint32_t)(first[base_line + (uint64_t)j] - F0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    auto slot_base = [&](uint32_t line_ord) -> uint32_t {         // kFill: first slot of a line, relative to F0
        return line_ord < kStageLines ? st_f[line_ord] : (uint32_t)(first[base_line + line_ord] - F0);
    };
    // kFirst: (line ordinal, start, end); kCount: (line ordinal, count, -); kFill: (slot relative to F0, start, end)
    // staged packed as start | end << 16 when both fit 16 bits (kNone stays kNone); anything else goes to memory directly
    uint32_t win_lo = 0;                                          // MULTI: first ordinal of the window being staged (idx is relative to it; wave-uniform)
    auto emit = [&](uint32_t idx, uint32_t v0, uint32_t v1) {
        const bool staged = idx < kStageLines;
        if (MODE == kCount) { if (staged) st_s[idx] = v0; else match_start[base_line + idx] = v0; return; }
        const bool fits = v0 == kNone || v1 < (MODE == kFirst ? 0x8000u : 0xffffu);      // (v0 <= v1; kFirst: bit 31 of a staged entry marks a PARKED hit)
        if (staged) st_s[idx] = v0 == kNone ? kNone : fits ? (v0 | v1 << 16) : kDirect;
        if (!staged || !fits) {
            const uint64_t at = ((MODE == kFill || MODE == kAll) ? F0 : base_line + win_lo) + idx;
            if (MODE != kAll || at < all.cap) { match_start[at] = v0; match_end[at] = v1; }
        }
    };
    uint32_t lane_base = 0;                                       // kAll: matches of the lanes before me in this chunk
    auto set_first = [&](uint32_t line_ord, uint32_t rel) {      // kAll: slot of a line's first match, relative to F0
        if (line_ord < kStageLines) st_f[line_ord] = rel; else all.first_out[base_line + line_ord] = F0 + rel;
    };
    // ---- 4. the events in byte order, then the rest of my last line beyond my bytes.  A hit whose start is not known QUEUES a job (lower
    // bound, match end, where to put the result, line start): walking back right at the hit would make the lanes of a
    // wave take turns, each waiting through the others' walks (measured: 33 ms for 8 GiB).  Positions are 32-bit
    // offsets from the chunk start from here on.
    // The whole of step 4 is one pass; kAll runs it twice: counting (nothing written, no walks), then placing.
    const uint32_t e_fwd = e, ord0 = ord;
    uint32_t emitted = 0;                                         // kFill / kAll: matches of lines that are mine
    // kAll: the hits the counting pass meets while following my last line beyond my bytes (position << 2 | flags), so
    // that the placing pass need not walk that stretch of text again; more than kFollowHits of them: it walks
    constexpr int kFollowHits = 4;
    uint32_t fh[kFollowHits];
    uint32_t nfh = 0;
    uint32_t wave_matches = 0;                                    // kAll: matches of the chunk (known between its two passes)
    auto run_pass = [&](auto counting_tag) {
    constexpr bool COUNTING = decltype(counting_tag)::value;
    // (r4) kAll, placing pass, when the chunk's matches all have a staged entry: a match is only PARKED at its slot by its own
    // lane (anchored << 31 | line start << 17 | match end relative to the line start) in a loop of the lane's own - no ballot, no
    // reservation, no walk job - and the slots are then taken 64 at a time, a match per lane (place_by_slot, behind the follow loop):
    // the lower bound of a match is the end of the match in the slot before if that is of the same line; an anchored match becomes
    // its result, the others are pooled for the walk.
    bool fast_place = false;
    // (the lines need not all be staged: the first slot of a line beyond the array goes to memory directly, as in the other path)
    // A match whose slot lies beyond the array is placed by its lane alone, there and then (a chunk with a match per byte: 16384).
    if constexpr (MODE == kAll && !COUNTING) fast_place = true;
    const bool all_staged = __builtin_amdgcn_readfirstlane((int)(wave_matches <= kStageLines)) != 0;       // (kAll; wave-uniform: a scalar)
    e = e_fwd; ord = ord0; emitted = 0;
    bool owned = fresh, decided = false;                          // the current line: is it mine; kFirst: has its match been found
    bool fast_count_pass = false;                                 // kCount: the hits added themselves to their lines' staged counts
    bool parked = false;                                          // kFirst: the hits of my bytes are parked in the staging array (wave-uniform)
    bool fh_valid = false;                                        // MULTI: the hit the follow loop found for my open last line
    uint32_t fh_pos = 0, fh_anchored = 0, fh_ord = 0, fh_ls = 0;
    if constexpr (MODE == kAll && !COUNTING) { if (fresh) set_first(ord0, lane_base); }
    const uint32_t my_rel = (uint32_t)(my - cstart), my_end_rel = (uint32_t)(my_end - cstart);
    uint32_t ls = my_rel;                                         // its first byte (valid if owned)
    uint32_t lb = my_rel;                                         // kFill: a match may not start before here (the previous match's end)
    uint32_t cnt = 0;                                             // kCount / kFill: matches of the current line so far
    uint32_t next_slot = kPool;                                   // where my next job goes in the pool (set by the caller of on_hit, wave-wide: see reserve)
    auto on_hit = [&](uint32_t pos, uint32_t f) {                 // the byte at `pos` completes a match
        if (!owned) return;
        if constexpr (MODE == kCount) { cnt++; return; }
        if constexpr (COUNTING) { emitted++; return; }
        if constexpr (MULTI) {                                    // (only the follow loop comes here: the hit of my open last line, kept for its window)
            fh_valid = true; fh_pos = pos; fh_anchored = f & 1u; fh_ord = ord; fh_ls = ls;
            decided = true;
            return;
        }
        const uint32_t at = MODE == kFirst ? ord : MODE == kAll ? lane_base + emitted : slot_base(ord) + cnt;
        const uint32_t lower = MODE == kFirst ? ls : lb;
        if (MODE == kAll && fast_place && at < kStageLines && pos + 1 - ls < 0x1fff0u) {
            st_s[at] = (f & 1u) << 31 | ls << 17 | (pos + 1 - ls);
            cnt++; emitted++; lb = pos + 1;
            return;
        }
        if (MODE == kAll && fast_place) {                         // beyond the array, or a match that ends 128 KiB into its line: placed here and now, alone
            const size_t s0 = f == 3u ? cstart + lower : reverse_walk(t, bytes, cstart + lower, cstart + pos + 1);
            emit(at, (uint32_t)(s0 - cstart) - ls, pos + 1 - ls);
            if (at < kStageLines) st_s[at] = kDirect;             // (not a parked record: place_by_slot leaves it alone)
            cnt++; emitted++; lb = pos + 1;
            return;
        }
        if (f == 3u) emit(at, lower - ls, pos + 1 - ls);          // accepted from the restart point itself: it starts there
        else {
            const uint32_t idx = next_slot++;
            if (idx < kPool) { pool[4 * idx] = lower; pool[4 * idx + 1] = pos + 1; pool[4 * idx + 2] = at; pool[4 * idx + 3] = ls; }
            else {                                                // (no room even after a drain: more jobs in one turn than the pool holds; the last bytes of the corpus)
                const size_t s0 = reverse_walk(t, bytes, cstart + lower, cstart + pos + 1);
                emit(at, (uint32_t)(s0 - cstart) - ls, pos + 1 - ls);
            }
        }
        if constexpr (MODE == kFirst) decided = true;
        else { cnt++; emitted++; lb = pos + 1; }
    };
    auto on_newline = [&](uint32_t pos) {
        if constexpr (MODE == kFirst && !MULTI) { if (owned && !decided && ord >= kStageLines) emit(ord, kNone, kNone); }      // (staged lines default to "none")
        if constexpr (MODE == kCount) { if (owned) emit(ord, cnt, 0u); }
        ord++;
        ls = pos + 1; lb = ls; cnt = 0;
        owned = ls < my_end_rel;                                  // a line that starts at my_end is the next lane's
        decided = false;
        if constexpr (MODE == kAll && !COUNTING) { if (owned) set_first(ord, lane_base + emitted); }
    };
    // ---- walks back.  A hit whose start is not known leaves a job (lower bound, match end, where the result goes, line
    // start) in the wave's POOL in LDS; whenever the pool holds more than 64 the wave walks 64 of them, a job per lane, in ONE
    // tight loop until the longest is done, then the lanes emit.  (Round 2 and the first form of round 3 kept a queue of four
    // jobs per lane in registers: a round of walks then served one job of every lane that had any, and the number of rounds
    // was the number of hits of the busiest lane - 12 on the email config where the mean is 6.5.)  A turn of the loop takes
    // the FOUR bytes below the walk's position: the text is re-read from L1/L2 as aligned words, a word requested a turn
    // ahead, and put together with v_alignbyte (the walk starts anywhere in a word); the four class lookups are issued before
    // the four dependent row lookups; the byte selects are static.  A lane that has died or passed its lower bound keeps
    // stepping to the end of the turn: row 0 leads to row 0, and a start below the bound is never taken.  8 VALU + 2 LDS per
    // reverse step (round 2: a byte per turn, a ballot and a branch per byte).  Positions are offsets from the chunk start.
    const uint8_t *const cbase = bytes + cstart;
    auto text_word = [&](uint32_t at) -> uint32_t { return *reinterpret_cast<const uint32_t *>(cbase + at); };   // at: multiple of 4
    uint32_t fill = 0;                                            // jobs in the pool (wave-uniform; the pool is empty between chunks and passes)
    // everything: every job in the pool; otherwise whole rounds of 64 only (the rest moves to the front and waits for company)
    auto drain = [&](bool everything) {
        const uint32_t count = fill;
        uint32_t base = 0;
        while (everything ? base < count : base + 64u <= count) {
            const uint32_t jn = base + (uint32_t)lane;
            const bool have = jn < count;
            base += 64u;
            uint32_t lo = 0, e_rel = 0, cur_at = 0, cur_ls = 0, kb = 0, best = 0, r = 0, hi_w = 0, lo_w = 0, sh = 0;
            if (have) {
                lo = pool[4 * jn]; e_rel = pool[4 * jn + 1]; cur_at = pool[4 * jn + 2]; cur_ls = pool[4 * jn + 3];
                kb = e_rel; best = e_rel; r = t.start_row;      // e_rel > lo: a match is never empty here
                sh = kb & 3u;
                const uint32_t a1 = kb & ~3u;                     // the word that holds byte kb: its low `sh` bytes are wanted
                if (sh) hi_w = text_word(a1);
                if (a1 >= 4u) lo_w = text_word(a1 - 4u);          // (a1 < 4: sh > 0 and lo = 0 - the wanted bytes are all in hi_w)
            }
            bool active = have;
            while (__ballot(active)) {
                if (active) {
                    // bytes kb-4 .. kb-1 = ({hi_w, lo_w} >> 8 sh): byte 3 of w is the nearest
                    const uint32_t w = __builtin_amdgcn_alignbyte(hi_w, lo_w, sh);
                    hi_w = lo_w;
                    const uint32_t below = (kb & ~3u) - 8u;       // requested now, used next turn
                    if (kb >= 8u) lo_w = text_word(below);
                    uint32_t a3, a2, a1_, a0;
                    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a3) : "v"(t.cls2_base), "v"(w));
                    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a2) : "v"(t.cls2_base), "v"(w));
                    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a1_) : "v"(t.cls2_base), "v"(w));
                    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a0) : "v"(t.cls2_base), "v"(w));
                    const uint32_t c3 = *reinterpret_cast<lds_u8_ptr>(a3), c2 = *reinterpret_cast<lds_u8_ptr>(a2);
                    const uint32_t c1 = *reinterpret_cast<lds_u8_ptr>(a1_), c0 = *reinterpret_cast<lds_u8_ptr>(a0);
                    const uint32_t room = kb - lo;                // steps this walk may still take (>= 1)
                    r = reverse_step(t, r, c3); if ((r & 1u) && room > 0u) best = kb - 1u;
                    r = reverse_step(t, r, c2); if ((r & 1u) && room > 1u) best = kb - 2u;
                    r = reverse_step(t, r, c1); if ((r & 1u) && room > 2u) best = kb - 3u;
                    r = reverse_step(t, r, c0); if ((r & 1u) && room > 3u) best = kb - 4u;
                    active = r != 0u && room > 4u;                // row 0: dead
                    kb -= 4u;                                     // (only read again while active: then kb > lo >= 0)
                }
            }
            if (have) emit(cur_at, best - cur_ls, e_rel - cur_ls);
        }
        const uint32_t rem = base < count ? count - base : 0u;    // < 64
        if (rem && base && (uint32_t)lane < rem) {
            const uint32_t j = base + (uint32_t)lane;
            const uint32_t v0 = pool[4 * j], v1 = pool[4 * j + 1], v2 = pool[4 * j + 2], v3 = pool[4 * j + 3];
            pool[4 * lane] = v0; pool[4 * lane + 1] = v1; pool[4 * lane + 2] = v2; pool[4 * lane + 3] = v3;
        }
        fill = (uint32_t)__builtin_amdgcn_readfirstlane((int)rem);
    };
    // Wave-uniform, before a turn in which the lanes push n jobs between them (exactly n: every reserved slot gets its job,
    // the pool has no holes): -> the first of their slots; slots >= kPool do not exist, those lanes walk alone.  The lanes
    // take consecutive slots in lane order (lane_offset: jobs of the lanes below me, from per-lane counts of 0 .. 4).
    auto reserve = [&](uint32_t n) -> uint32_t {
        if (fill + n > kPool) { drain(false); if (fill + n > kPool) drain(true); }
        const uint32_t first_slot = fill;
        fill = (uint32_t)__builtin_amdgcn_readfirstlane((int)(fill + n < kPool ? fill + n : kPool));      // (wave-uniform: keep it in a scalar register)
        return first_slot;
    };
    auto below = [&](uint64_t m) -> uint32_t { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
    auto reserve_one = [&](bool push) {                           // I am about to push a job, or not; sets next_slot
        const uint64_t m1 = __ballot(push);
        if (m1) { const uint32_t first_slot = reserve((uint32_t)__popcll(m1)); if (push) next_slot = first_slot + below(m1); }
    };
    auto reserve_for = [&](uint32_t mine) {                       // mine: 0 .. 4 jobs I am about to push; sets next_slot
        const uint64_t m1 = __ballot(mine > 0u);
        if (!m1) return;
        uint32_t n = (uint32_t)__popcll(m1), off = below(m1);
        if constexpr (MODE != kFirst) {
            const uint64_t m2 = __ballot(mine > 1u), m3 = __ballot(mine > 2u), m4 = __ballot(mine > 3u);
            n += (uint32_t)__popcll(m2) + (uint32_t)__popcll(m3) + (uint32_t)__popcll(m4);
            off += below(m2) + below(m3) + below(m4);
        }
        next_slot = reserve(n) + off;
    };
    constexpr bool kWalks = !COUNTING && MODE != kCount;          // this pass queues jobs
    constexpr int kFollowTurnWords = (kWalks && MODE != kFirst) ? 1 : 4;      // text words per turn of the follow loop (see there)
    // ---- 4a. my events, word by word (a static loop over the event words, a dynamic one over the events of a word: the
    // version with one loop body that rotated the words through ev[0] paid 7 extra turns per lane and a longer body)
    if constexpr (MODE == kAll && COUNTING) {
        // counting needs no replay: every hit among my events is mine (a lane that begins inside a line sits in the SKIP
        // row until its first '\n', so it records no hit for that line), and my last line is mine if any line starts here
#pragma unroll
        for (int i = 0; i < kEv; i++) emitted += __popc(ev[i] & 0xaaaaaaaau);
        owned = fresh || nl > 0;
    } else {
    bool fast_first = false, fast_count = false;
    if constexpr (MODE == kFirst) fast_first = MULTI || hi_ord < (int64_t)kStageLines;     // every line of the chunk has a staged entry ("none" by default; MULTI: window by window)
    if constexpr (MODE == kCount) fast_count = hi_ord < (int64_t)kStageLines;      // ... (0 by default)
    if (fast_count) {
        fast_count_pass = true;
        // matches per line, the common case: only the hits are visited here too - each adds one to its line's staged count (a
        // line's hits are all its owner's: the lanes behind sit in SKIP until its '\n'), the '\n' events are only counted
        uint32_t run_ord = ord0, run_ls = my_rel;
#pragma unroll
        for (int i = 0; i < kEv; i++) {
            const uint32_t evw = ev[i];
            const uint32_t nlm = evw & ~(evw >> 1) & 0x55555555u;
            uint32_t hm = evw & 0xaaaaaaaau;
            while (hm) {
                const int zb = __clz((int)hm);
                hm &= ~(0x80000000u >> zb);
                const uint32_t above = zb ? nlm & ~(0xffffffffu >> zb) : 0u;
                atomicAdd(&st_s[run_ord + (uint32_t)__popc(above)], 1u);
            }
            if (nlm) { run_ord += (uint32_t)__popc(nlm); run_ls = my_rel + (uint32_t)(16 * i) + ((uint32_t)(31 - __ffs((int)nlm)) >> 1) + 1u; }
        }
        ord = run_ord; ls = run_ls; lb = run_ls; cnt = 0;             // (cnt: what the follow loop adds to my open last line)
        owned = nl ? run_ls < my_end_rel : fresh;
    } else if (fast_first) {
        // first match per line, the common case: only the HITS are visited.  The forward table leaves one hit per line (SKIP
        // until the '\n'), every hit among my events is mine (see above), and what a hit needs of the '\n' events - its line's
        // ordinal and first byte - are a popcount and a find-first over the '\n' fields in front of it.  (Round 2 took every
        // event in turn, '\n' and hit alike, each a pass through the whole body: on 20-byte lines the event loop cost twice the
        // forward pass; on 5-byte lines three quarters of the events are '\n'.)
        // (r4) A hit is only PARKED here - at its line's ordinal in the staging array: bit 31 | anchored << 30 | line start << 16 |
        // match end relative to the line start (both inside the chunk: 14 and 15 bits) - by its own lane in its own loop: no ballot, no
        // reservation, no walk job.  The lines are then taken 64 at a time, a line per lane (parked_lines, behind the follow loop):
        // an anchored hit becomes its result, the others are pooled for the walk with every lane of the wave in the game.  (The
        // round-3 loop took a hit of every lane per turn of a wave-wide loop and did all of that inside it: 0.20 of the email
        // config's 0.64 ms per GiB, 0.67 of the URL config's 3.2 ms per 8 GiB with the write-out.)
        uint32_t run_ord = ord0, run_ls = my_rel, last_hit_end = 0;
#pragma unroll
        for (int i = 0; i < kEv; i++) {
            const uint32_t evw = ev[i];
            const uint32_t nlm = evw & ~(evw >> 1) & 0x55555555u;     // fields equal to 1: bit 30 - 2 y for byte y
            uint32_t hm = evw & 0xaaaaaaaau;                          // fields 2 and 3: bit 31 - 2 y
            while (hm) {
                const int zb = __clz((int)hm);                        // 2 y
                hm &= ~(0x80000000u >> zb);
                const uint32_t anchored = (evw >> (30 - zb)) & 1u;
                const uint32_t above = zb ? nlm & ~(0xffffffffu >> zb) : 0u;          // the '\n' of this word in front of the hit
                const uint32_t pos = my_rel + (uint32_t)(16 * i) + (uint32_t)(zb >> 1);
                const uint32_t o = run_ord + (uint32_t)__popc(above);
                const uint32_t l = above ? my_rel + (uint32_t)(16 * i) + ((uint32_t)(31 - __ffs((int)above)) >> 1) + 1u : run_ls;
                last_hit_end = pos + 1u;
                if (!MULTI || o < kStageLines) st_s[o] = 0x80000000u | anchored << 30 | l << 16 | (pos + 1u - l);       // (MULTI: window 0 here, the others behind the follow loop)
            }
            if (nlm) { run_ord += (uint32_t)__popc(nlm); run_ls = my_rel + (uint32_t)(16 * i) + ((uint32_t)(31 - __ffs((int)nlm)) >> 1) + 1u; }
        }
        parked = true;
        ord = run_ord; ls = run_ls; lb = run_ls; cnt = 0;
        owned = nl ? run_ls < my_end_rel : fresh;                     // a line that starts at my_end is the next lane's
        decided = last_hit_end > run_ls;                              // my open last line has its match already
    } else if (fast_place) {
        // every event in byte order, in the lane's own loop: a '\n' sets the next line's first slot, a hit is parked at its slot (on_hit)
#pragma unroll
        for (int i = 0; i < kEv; i++) {
            uint32_t m = ev[i];
            while (m) {
                const int z = __clz((int)m) >> 1;
                const uint32_t f = (m >> (30 - 2 * z)) & 3u;
                m &= ~(3u << (30 - 2 * z));
                const uint32_t pos = my_rel + (uint32_t)(16 * i + z);
                if (f == 1u) on_newline(pos);
                else if (!all_staged) on_hit(pos, f);
                else if (owned) {                                 // (every match of the chunk has a slot in the array, and inside my own bytes a match ends
                                                                  //  less than 16 KiB into its line: parked without further ado)
                    st_s[lane_base + emitted] = (f & 1u) << 31 | ls << 17 | (pos + 1 - ls);
                    cnt++; emitted++; lb = pos + 1;
                }
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < kEv; i++) {
        uint32_t m = ev[i];
        if constexpr (kWalks) {
            while (__ballot(m != 0)) {                            // (a wave-uniform loop: the walks in between need every lane)
                const uint32_t z = (uint32_t)__clz((int)m) >> 1;  // byte of the word, 0 = first (no event left: 16, f = 0)
                const uint32_t sh = (30u - 2u * z) & 31u;
                const uint32_t f = m ? (m >> sh) & 3u : 0u;
                m &= ~(3u << sh);
                const uint32_t pos = my_rel + (uint32_t)(16 * i) + z;
                reserve_one(f == 2u && owned);
                if (f == 1u) on_newline(pos); else if (f) on_hit(pos, f);
            }
        } else {
            while (m) {
                const int z = __clz((int)m) >> 1;
                const uint32_t f = (m >> (30 - 2 * z)) & 3u;
                m &= ~(3u << (30 - 2 * z));
                const uint32_t pos = my_rel + (uint32_t)(16 * i + z);
                if (f == 1u) on_newline(pos); else on_hit(pos, f);
            }
        }
    }
    }
    }
    // my last line goes on beyond my bytes and is still open (kFirst: undecided): follow it
    int phase = (vlen == kSearchS && !my_last_nl && owned && !(MODE == kFirst && decided)) ? 1 : 2;   // 1: following, 2: done
    auto follow_hit = [&](uint32_t pos, uint32_t f) {
        if constexpr (MODE == kAll && COUNTING) {
#pragma unroll
            for (int j = 0; j < kFollowHits; j++)
                if ((uint32_t)j == nfh) fh[j] = pos << 2 | f;
            nfh = pos < (1u << 30) ? nfh + 1 : (uint32_t)kFollowHits + 1;      // (a position that does not fit: walk again)
        }
        on_hit(pos, f);
    };
    if constexpr (MODE == kAll && !COUNTING) {
        {
            uint32_t mine = 0;                                    // the remembered hits whose start is not known
            if (phase == 1 && nfh <= (uint32_t)kFollowHits) {
#pragma unroll
                for (int j = 0; j < kFollowHits; j++)
                    if ((uint32_t)j < nfh && (fh[j] & 3u) == 2u) mine++;
            }
            if (!fast_place) reserve_for(mine);
        }
        if (phase == 1 && nfh <= (uint32_t)kFollowHits) {         // the counting pass has been there: its hits, in order
#pragma unroll
            for (int j = 0; j < kFollowHits; j++)
                if ((uint32_t)j < nfh) on_hit(fh[j] >> 2, fh[j] & 3u);
            phase = 2;
        }
    }
    size_t fbyte = my_end;                                        // follow position: a multiple of 16 (my_end is a multiple of kSearchS, or the end of the data)
    uint4 cur = make_uint4(0, 0, 0, 0), nxt = cur;
    if (phase == 1 && fbyte + 16 <= nbytes) cur = *reinterpret_cast<const uint4 *>(bytes + fbyte);
    {
        // ---- 4b. the rest of my last line: SIXTEEN bytes per turn (round 3: four - the loop runs until the wave's longest line
        // is done, and what a turn costs beside its steps was paid per four bytes: 0.98 of 3.69 ms on the URL config,
        // profiles/r04_search_ablation.txt), the next sixteen requested before these are stepped (the loop is a chain of memory
        // round trips otherwise), stepped like the forward pass
        // kTurnWords text words per turn: four, or one in the passes that place every match (kFill, the second pass of kAll:
        // where every byte is a match - all matches of a{1,300} - the sixteen-byte turn measured 38 % SLOWER, 22.5 against 16.3 ms
        // per GiB in the fill pass; tools/probe/search_ablate/follow_ab.sh)
        constexpr int kTurnWords = kFollowTurnWords;
        auto follow_step = [&](uint32_t &acc, uint32_t &wrel) {   // -> the events of the bytes stepped (2 bits per byte, the first byte in bits 31..30)
            const size_t fpos = fbyte & ~(size_t)15;
            if (fpos + 16 <= nbytes) {
                const uint32_t fq = (uint32_t)(fbyte & 15) >> 2;  // text word of `cur` to step (four words per turn: always 0)
                if (fq == 0 && fpos + 32 <= nbytes) nxt = *reinterpret_cast<const uint4 *>(bytes + fpos + 16);
                if constexpr (kTurnWords == 4) {
                    uint4 v = cur;
                    if (clean) { v.x = clean_word(v.x); v.y = clean_word(v.y); v.z = clean_word(v.z); v.w = clean_word(v.w); }
                    {
                        uint32_t i0, i1, i2, i3;
                        pair_index(v.x, i0, i1); pair_index(v.y, i2, i3);
                        const uint32_t c0 = pair_column(i0), c1 = pair_column(i1), c2 = pair_column(i2), c3 = pair_column(i3);
                        step_pair(e, c0, acc); step_pair(e, c1, acc); step_pair(e, c2, acc); step_pair(e, c3, acc);
                    }
                    {
                        uint32_t i0, i1, i2, i3;
                        pair_index(v.z, i0, i1); pair_index(v.w, i2, i3);
                        const uint32_t c0 = pair_column(i0), c1 = pair_column(i1), c2 = pair_column(i2), c3 = pair_column(i3);
                        step_pair(e, c0, acc); step_pair(e, c1, acc); step_pair(e, c2, acc); step_pair(e, c3, acc);
                    }
                } else {
                    uint32_t w = fq == 0 ? cur.x : fq == 1 ? cur.y : fq == 2 ? cur.z : cur.w;
                    if (clean) w = clean_word(w);
                    uint32_t ia, ib;
                    pair_index(w, ia, ib);
                    const uint32_t ca = pair_column(ia), cb = pair_column(ib);
                    step_pair(e, ca, acc); step_pair(e, cb, acc);
                    acc <<= 24;
                }
                wrel = (uint32_t)(fbyte - cstart);
                fbyte += 4 * kTurnWords;
                if ((fbyte & 15) == 0) cur = nxt;
            } else {                                              // the last bytes of the corpus, pair by pair
                next_slot = kPool;                                // (no slots reserved here: these few hits walk alone)
                for (; fbyte < nbytes && phase == 1; fbyte += 2) {  // (fbyte is even: pairs stay aligned; a last odd byte is paired with 0x00)
                    uint32_t c1 = bytes[fbyte], c2 = fbyte + 1 < nbytes ? bytes[fbyte + 1] : 0u;
                    if (c1 & 0x80u) c1 = 0;
                    if (c2 & 0x80u) c2 = 0;
                    uint32_t pa = 0;
                    {
                        uint32_t ia, ib;
                        pair_index(c1 | c2 << 8, ia, ib);
                        step_pair(e, pair_column(ia), pa);
                    }
                    const uint32_t f1 = (pa >> 2) & 3u, f2 = pa & 3u;
                    if (f1 == 1u) phase = 2;
                    else if (f1) { follow_hit((uint32_t)(fbyte - cstart), f1); if (MODE == kFirst) phase = 2; }
                    if (phase == 1 && fbyte + 1 < nbytes) {
                        if (f2 == 1u) phase = 2;
                        else if (f2) { follow_hit((uint32_t)(fbyte + 1 - cstart), f2); if (MODE == kFirst) phase = 2; }
                    }
                }
                if (fbyte >= nbytes) phase = 2;                   // the end of the data ends the line
            }
        };
        auto follow_events = [&](uint32_t acc, uint32_t wrel) {   // the events of a turn (or of a part of it: fields masked out are zero) in byte order
            while (acc && phase == 1) {
                const int z = __clz((int)acc) >> 1;
                const uint32_t f = (acc >> (30 - 2 * z)) & 3u;
                acc &= ~(3u << (30 - 2 * z));
                if (f == 1u) phase = 2;                           // the line's '\n': done (what lies behind it is not mine)
                else { follow_hit(wrel + (uint32_t)z, f); if (MODE == kFirst) phase = 2; }
            }
        };
        if constexpr (kWalks) {
            // wave-uniform, like the event loop: the walks in between need every lane
            while (__ballot(phase == 1)) {
                uint32_t acc = 0, wrel = 0;
                if (phase == 1) follow_step(acc, wrel);
                if (__ballot(acc != 0)) {                         // (most turns meet neither a '\n' nor a hit)
                    // only the events up to the one that ends my following count: the line's '\n' (kFirst: or the first hit)
                    uint32_t stop = acc & ~(acc >> 1) & 0x55555555u;
                    if constexpr (MODE == kFirst) stop |= (acc >> 1) & 0x55555555u;
                    if (stop) acc &= ~((1u << (31 - __clz((int)stop))) - 1u);
                    // (first match: at most one hit is left; one word per turn: at most four; MULTI keeps the hit for its window: no job here)
                    if constexpr (!MULTI) { if (!fast_place) reserve_for((uint32_t)__popc((acc >> 1) & ~acc & 0x55555555u)); }     // my hits whose start is not known (fields equal to 2)
                    follow_events(acc, wrel);
                }
            }
        } else {
            while (phase == 1) {
                uint32_t acc = 0, wrel = 0;
                follow_step(acc, wrel);
                follow_events(acc, wrel);
            }
        }
        // ---- 4b''. MULTI: the same window by window; every window is written out here
        if constexpr (MULTI) {
            auto park_window = [&]() {                                                // my hits whose ordinal lies in the window (as 4a did for window 0)
                uint32_t run_ord = ord0, run_ls = my_rel;
#pragma unroll
                for (int i = 0; i < kEv; i++) {
                    const uint32_t evw = ev[i];
                    const uint32_t nlm = evw & ~(evw >> 1) & 0x55555555u;
                    uint32_t hm = evw & 0xaaaaaaaau;
                    while (hm) {
                        const int zb = __clz((int)hm);
                        hm &= ~(0x80000000u >> zb);
                        const uint32_t anchored = (evw >> (30 - zb)) & 1u;
                        const uint32_t above = zb ? nlm & ~(0xffffffffu >> zb) : 0u;
                        const uint32_t pos = my_rel + (uint32_t)(16 * i) + (uint32_t)(zb >> 1);
                        const uint32_t o = run_ord + (uint32_t)__popc(above) - win_lo;
                        const uint32_t l = above ? my_rel + (uint32_t)(16 * i) + ((uint32_t)(31 - __ffs((int)above)) >> 1) + 1u : run_ls;
                        if (o < kStageLines) st_s[o] = 0x80000000u | anchored << 30 | l << 16 | (pos + 1u - l);
                    }
                    if (nlm) { run_ord += (uint32_t)__popc(nlm); run_ls = my_rel + (uint32_t)(16 * i) + ((uint32_t)(31 - __ffs((int)nlm)) >> 1) + 1u; }
                }
            };
            for (win_lo = 0;; win_lo += kStageLines) {                                // (wave-uniform)
                if (win_lo) park_window();
                // the hit the follow loop found for my open last line, if that line is in this window
                if (fh_valid && fh_ord - win_lo < kStageLines) {
                    const uint32_t rel = fh_ord - win_lo, endrel = fh_pos + 1u - fh_ls;
                    if (endrel < 0x8000u) st_s[rel] = 0x80000000u | fh_anchored << 30 | fh_ls << 16 | endrel;
                    else if (fh_anchored) emit(rel, 0u, endrel);                      // (a match that ends 32 KiB into its line: to memory directly)
                    else {
                        const size_t s0 = reverse_walk(t, bytes, cstart + fh_ls, cstart + fh_pos + 1);
                        emit(rel, (uint32_t)(s0 - cstart) - fh_ls, endrel);
                    }
                }
                const int64_t first_rel = (int64_t)lo_ord > (int64_t)win_lo ? (int64_t)lo_ord - win_lo : 0;
                const int64_t last_rel = hi_ord - (int64_t)win_lo < (int64_t)kStageLines - 1 ? hi_ord - (int64_t)win_lo : (int64_t)kStageLines - 1;
                for (int64_t j0 = first_rel; j0 <= last_rel; j0 += 64) {              // (wave-uniform)
                    const int64_t j = j0 + lane;
                    const uint32_t v = j <= last_rel ? st_s[j] : kNone;
                    const bool is_parked = v != kNone && v != kDirect && (v >> 31) != 0u;
                    const uint32_t l = (v >> 16) & 0x3fffu, endrel = v & 0xffffu;
                    const bool push = is_parked && !((v >> 30) & 1u);
                    if (is_parked && !push) st_s[j] = endrel << 16;                   // anchored: the match is [0, endrel) of its line
                    reserve_one(push);
                    if (push) {
                        const uint32_t idx = next_slot;
                        if (idx < kPool) { pool[4 * idx] = l; pool[4 * idx + 1] = l + endrel; pool[4 * idx + 2] = (uint32_t)j; pool[4 * idx + 3] = l; }
                        else {                                                        // (no room even after a drain)
                            const size_t s0 = reverse_walk(t, bytes, cstart + l, cstart + l + endrel);
                            emit((uint32_t)j, (uint32_t)(s0 - cstart) - l, endrel);
                        }
                    }
                }
                drain(true);
                // the window's results: whole sectors, consecutive lanes consecutive lines; the entries back to "none" for the next window / chunk
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int64_t j = first_rel + lane; j <= last_rel; j += 64) {
                    const uint32_t v = st_s[j];
                    if (v != kDirect) {
                        match_start[base_line + win_lo + (uint64_t)j] = v == kNone ? kNone : (v & 0xffffu);
                        match_end[base_line + win_lo + (uint64_t)j] = v == kNone ? kNone : (v >> 16);
                    }
                    st_s[j] = kNone;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if ((int64_t)win_lo + (int64_t)kStageLines > hi_ord) break;
            }
            win_lo = 0;
        }
        // ---- 4b'. kFirst: the parked hits, a LINE per lane
        if constexpr (MODE == kFirst && !MULTI) {
            if (parked) {
                for (int64_t j0 = (int64_t)lo_ord; j0 <= hi_ord; j0 += 64) {          // (wave-uniform; every ordinal of the chunk is staged here)
                    const int64_t j = j0 + lane;
                    const uint32_t v = j <= hi_ord ? st_s[j] : kNone;
                    const bool is_parked = v != kNone && v != kDirect && (v >> 31) != 0u;
                    const uint32_t l = (v >> 16) & 0x3fffu, endrel = v & 0xffffu;
                    const bool push = is_parked && !((v >> 30) & 1u);
                    if (is_parked && !push) st_s[j] = endrel << 16;                   // anchored: the match is [0, endrel) of its line
                    reserve_one(push);
                    if (push) {
                        const uint32_t idx = next_slot;
                        if (idx < kPool) { pool[4 * idx] = l; pool[4 * idx + 1] = l + endrel; pool[4 * idx + 2] = (uint32_t)j; pool[4 * idx + 3] = l; }
                        else {                                                        // (no room even after a drain)
                            const size_t s0 = reverse_walk(t, bytes, cstart + l, cstart + l + endrel);
                            emit((uint32_t)j, (uint32_t)(s0 - cstart) - l, endrel);
                        }
                    }
                }
            }
        }
        // ---- 4b3. kAll: the parked matches, a MATCH per lane (place_by_slot)
        if constexpr (MODE == kAll && !COUNTING) {
            if (fast_place) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                uint32_t carry = kDirect;                                             // the record of the slot before this round's first (none yet)
                const uint32_t staged_matches = wave_matches < kStageLines ? wave_matches : kStageLines;
                for (uint32_t k0 = 0; k0 < staged_matches; k0 += 64) {                // (wave-uniform)
                    const uint32_t k = k0 + (uint32_t)lane;
                    const bool have = k < staged_matches;
                    const uint32_t rec = have ? st_s[k] : kDirect;
                    uint32_t prev = __shfl_up(rec, 1, 64);
                    if (lane == 0) prev = carry;
                    carry = __shfl(rec, 63, 64);
                    const bool is_rec = have && rec != kDirect;
                    const uint32_t l = (rec >> 17) & 0x3fffu, endrel = rec & 0x1ffffu;
                    // the lower bound of my match: the end of the match before it if that one is of the same line (same line start)
                    const uint32_t lb_rel = (prev != kDirect && ((prev >> 17) & 0x3fffu) == l) ? (prev & 0x1ffffu) : 0u;
                    const bool push = is_rec && !(rec >> 31);
                    reserve_one(push);
                    if (is_rec && !push) emit(k, lb_rel, endrel);                     // it starts at the restart point
                    if (push) {
                        const uint32_t idx = next_slot;
                        if (idx < kPool) { pool[4 * idx] = l + lb_rel; pool[4 * idx + 1] = l + endrel; pool[4 * idx + 2] = k; pool[4 * idx + 3] = l; }
                        else {                                                        // (no room even after a drain)
                            const size_t s0 = reverse_walk(t, bytes, cstart + l + lb_rel, cstart + l + endrel);
                            emit(k, (uint32_t)(s0 - cstart) - l, endrel);
                        }
                    }
                }
            }
        }
        // ---- 4c. the walks still waiting
        if constexpr (kWalks) drain(true);
    }
    // my last line, open to the end of the data (or ended by its '\n' beyond my bytes)
    if constexpr (MODE == kFirst) { if (owned && !decided && ord >= kStageLines) emit(ord, kNone, kNone); }
    if constexpr (MODE == kCount) { if (owned) { if (fast_count_pass) atomicAdd(&st_s[ord], cnt); else emit(ord, cnt, 0u); } }
    };      // run_pass
    if constexpr (MODE == kAll) {
        run_pass(std::true_type{});
        uint32_t incl_m = emitted;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl_m, d, 64); if (lane >= d) incl_m += o; }
        wave_matches = __shfl(incl_m, 63, 64);
        lane_base = incl_m - emitted;
        F0 = chunk_lookback(all.status, all.ticket, chunk, wave_matches, lane);
        run_pass(std::false_type{});
    } else {
        run_pass(std::false_type{});
    }
    // ---- 5. write the wave's results: whole sectors, consecutive lanes consecutive entries
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the staging stores of every lane of this wave are done
    if constexpr (MODE == kFill) {
        uint32_t tot = emitted;
#pragma unroll
        for (int d = 32; d; d >>= 1) tot += __shfl_xor(tot, d, 64);
        const uint32_t staged = tot < kStageLines ? tot : kStageLines;
        for (uint32_t j = lane; j < staged; j += 64) {
            const uint32_t v = st_s[j];
            if (v != kDirect) { match_start[F0 + j] = v & 0xffffu; match_end[F0 + j] = v >> 16; }
        }
    } else if constexpr (MODE == kAll) {
        const uint32_t staged = wave_matches < kStageLines ? wave_matches : kStageLines;
        for (uint32_t j = lane; j < staged; j += 64) {
            const uint32_t v = st_s[j];
            if (v != kDirect && F0 + j < all.cap) { match_start[F0 + j] = v & 0xffffu; match_end[F0 + j] = v >> 16; }
        }
        int64_t hi = hi_ord;
        if (hi >= (int64_t)kStageLines) hi = (int64_t)kStageLines - 1;
        for (int64_t j = lo_ord + lane; j <= hi; j += 64) all.first_out[base_line + (uint64_t)j] = F0 + st_f[j];
        if (cend == nbytes && lane == 0) { all.first_out[all.nlines] = F0 + wave_matches; *all.total = F0 + wave_matches; }
    } else if constexpr (!MULTI) {                                // (MULTI wrote its windows inside the pass)
        int64_t hi = hi_ord;
        if (hi >= (int64_t)kStageLines) hi = (int64_t)kStageLines - 1;
        for (int64_t j = lo_ord + lane; j <= hi; j += 64) {
            const uint32_t v = st_s[j];
            if (MODE == kCount) match_start[base_line + (uint64_t)j] = v;
            else if (v != kDirect) {
                match_start[base_line + (uint64_t)j] = v == kNone ? kNone : (v & 0xffffu);
                match_end[base_line + (uint64_t)j] = v == kNone ? kNone : (v >> 16);
            }
        }
        // the staging array back to its default for the wave's next chunk (only ordinals <= total_nl were touched)
        const uint32_t used = total_nl + 1 < kStageLines ? total_nl + 1 : kStageLines;
        for (uint32_t j = lane; j < used; j += 64) st_s[j] = kStageInit;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

}  // namespace

size_t search_chunk_bytes() { return kSearchChunk; }
// forward tables (LDS form: pair table, gap, table; global form: pair table) + reverse table + class map + the waves' job pools
static size_t search_table_bytes(const SearchChunkDevice &p) {
    const size_t fwd = p.in_global ? (size_t)kSearchP16Bytes : ((size_t)p.base_row + p.nrows) * p.row_bytes;
    return fwd + ((size_t)(p.nr * p.ncls + 1) / 2 + 64 + (size_t)kSearchWaves * kPoolWords) * 4;
}
// staged entries per wave: what the tables and the job pools leave of the budget (the whole LDS of a CU), a multiple of 64, 128 at least
// (one array of packed results or counts; the fill pass also keeps the lines' slot bases)
static uint32_t search_stage_lines(const SearchChunkDevice &p, int mode) {
    const size_t tb = search_table_bytes(p), per = (size_t)kSearchWaves * 4 * ((mode == kFill || mode == kAll) ? 2 : 1);
    if (tb + per * 128 > kSearchChunkLdsBudget) return 0;
    if ((size_t)p.nr * p.ncls * 2 > 65534 || p.ncls > 128) return 0;      // reverse rows are addressed by 16-bit byte offsets, classes doubled in a byte
    if (!p.in_global && (p.base_row + p.nrows > 4096 || p.ncols2 > 127 || (p.row_bytes & 3) || p.base_row * p.row_bytes < kSearchP8Bytes)) return 0;
    if (p.in_global && ((size_t)p.nrows * p.ncols2 * 4 >= ((size_t)1 << 28) || p.ncols2 > 16383)) return 0;
    size_t n = (kSearchChunkLdsBudget - tb) / per;
    n = n / 64 * 64;
    return (uint32_t)(n > kMaxStageLines ? kMaxStageLines : n);
}
size_t search_chunks_lds_bytes(const SearchChunkDevice &p) {          // of the most demanding mode (fill)
    const uint32_t n = search_stage_lines(p, kFill);
    return n ? search_table_bytes(p) + (size_t)kSearchWaves * 8 * n : (size_t)kSearchChunkLdsBudget + 1;
}
template <int MODE, int FORM, bool MULTI = false>
static int launch_search_chunks_form(const SearchChunkDevice &p, bool clean, const uint8_t *bytes, size_t nbytes, const uint64_t *chunk_base, size_t nchunks,
                                     const uint64_t *first, uint32_t *out0, uint32_t *out1, void *stream, SearchAllArgs all) {
    if (!nchunks) return 0;
    const uint32_t lines = search_stage_lines(p, MODE);
    if (!lines) return (int)hipErrorInvalidValue;
    const size_t lds = search_table_bytes(p) + (size_t)kSearchWaves * 4 * ((MODE == kFill || MODE == kAll) ? 2 : 1) * lines;
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(search_chunks_kernel<MODE, FORM, MULTI>), lds, /*at_zero=*/true);
    if (e != hipSuccess) return (int)e;
    // persistent workgroups: the tables are loaded once per workgroup, its waves take chunk after chunk.  One workgroup per
    // CU and no more: a workgroup holds the CU's whole LDS, so a second generation could only start on a CU when all sixteen
    // waves of the first had finished (measured: four generations cost 8 % on the count pass).
    static std::atomic<int> cus_of[kMaxDevices];                  // (per device: 0 = not asked yet)
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess) dev_id = 0;
    int cus = dev_id >= 0 && dev_id < kMaxDevices ? cus_of[dev_id].load(std::memory_order_relaxed) : 0;
    if (!cus) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
        if (dev_id >= 0 && dev_id < kMaxDevices) cus_of[dev_id].store(cus, std::memory_order_relaxed);
    }
    size_t blocks = (nchunks + kSearchWaves - 1) / kSearchWaves;
    if (blocks > (size_t)cus) blocks = (size_t)cus;
    hipLaunchKernelGGL((search_chunks_kernel<MODE, FORM, MULTI>), dim3((unsigned)blocks), dim3(kSearchWaves * 64), lds, (hipStream_t)stream, p, clean ? 1u : 0u, bytes,
                       nbytes, chunk_base, nchunks, first, out0, out1, lines, all);
    return (int)hipGetLastError();
}
template <int MODE>
static int launch_search_chunks(const SearchChunkDevice &p, bool clean, const uint8_t *bytes, size_t nbytes, const uint64_t *chunk_base, size_t nchunks,
                                const uint64_t *first, uint32_t *out0, uint32_t *out1, void *stream, SearchAllArgs all = SearchAllArgs()) {
    return p.in_global ? launch_search_chunks_form<MODE, kGlobalForm>(p, clean, bytes, nbytes, chunk_base, nchunks, first, out0, out1, stream, all)
                       : launch_search_chunks_form<MODE, kLdsForm>(p, clean, bytes, nbytes, chunk_base, nchunks, first, out0, out1, stream, all);
}
int search_chunks(const SearchChunkDevice &p, bool clean, const uint8_t *bytes, size_t nbytes, const uint64_t *chunk_base, size_t nchunks, size_t nlines,
                  uint32_t *match_start, uint32_t *match_end, void *stream) {
    // a corpus whose chunks hold more lines than the staging array (on average, with a margin): the build that takes them in windows
    const uint32_t staged = search_stage_lines(p, kFirst);
    if (nchunks && staged && (double)nlines / (double)nchunks > 0.8 * staged) {
        SearchAllArgs none;
        return p.in_global ? launch_search_chunks_form<kFirst, kGlobalForm, true>(p, clean, bytes, nbytes, chunk_base, nchunks, nullptr, match_start, match_end, stream, none)
                           : launch_search_chunks_form<kFirst, kLdsForm, true>(p, clean, bytes, nbytes, chunk_base, nchunks, nullptr, match_start, match_end, stream, none);
    }
    return launch_search_chunks<kFirst>(p, clean, bytes, nbytes, chunk_base, nchunks, nullptr, match_start, match_end, stream);
}
int search_chunks_count(const SearchChunkDevice &p, bool clean, const uint8_t *bytes, size_t nbytes, const uint64_t *chunk_base, size_t nchunks,
                        uint32_t *count, void *stream) {
    return launch_search_chunks<kCount>(p, clean, bytes, nbytes, chunk_base, nchunks, nullptr, count, nullptr, stream);
}
int search_chunks_fill(const SearchChunkDevice &p, bool clean, const uint8_t *bytes, size_t nbytes, const uint64_t *chunk_base, size_t nchunks,
                       const uint64_t *first, uint32_t *match_start, uint32_t *match_end, void *stream) {
    return launch_search_chunks<kFill>(p, clean, bytes, nbytes, chunk_base, nchunks, first, match_start, match_end, stream);
}

// scratch of search_chunks_all: status u64[nchunks] | total u64 | ticket u32[2]; zeroed by the caller before every launch
size_t search_all_scratch_bytes(size_t nchunks) { return nchunks * sizeof(uint64_t) + 16; }
int search_chunks_all(const SearchChunkDevice &p, bool clean, const uint8_t *bytes, size_t nbytes, const uint64_t *chunk_base, size_t nchunks, size_t nlines,
                      uint64_t *first, uint32_t *match_start, uint32_t *match_end, size_t cap, void *scratch, void *stream) {
    SearchAllArgs a;
    a.first_out = first;
    a.status = static_cast<uint64_t *>(scratch);
    a.total = a.status + nchunks;
    a.ticket = reinterpret_cast<uint32_t *>(a.total + 1);
    a.cap = cap;
    a.nlines = nlines;
    return launch_search_chunks<kAll>(p, clean, bytes, nbytes, chunk_base, nchunks, nullptr, match_start, match_end, stream, a);
}

}  // namespace dev
}  // namespace rrx
