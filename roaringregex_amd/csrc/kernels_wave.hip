// kernels_wave.hip — the wave-resident NFA engine: ONE WAVE holds one state set of up to 65536 positions and steps one
// string.  Replaces round 2's block-cooperative engine (a workgroup per string, one barrier per byte: 4-5 GB/s).
//
// NFA.cc:77-85 restated for automata of any size the front end admits (Parser.cpp:165 takes any states_n):
//   * lane l holds WL consecutive 32-bit words of the set - positions [32 WL l, 32 WL (l + 1)) - in registers, WL = 1 ... 32;
//     the shift inside a lane is v_alignbit over its own words, the bit that crosses into the next lane arrives by DPP
//     wave_shr:1.  Nothing crosses waves: no barrier, no LDS exchange per byte;
//   * the text is the same for all 64 lanes: it is read 16 bytes at a time from a wave-uniform address (scalar loads), the
//     byte -> class -> B row address arithmetic is scalar;
//   * B rows per byte VALUE in HBM/L2, [byte][lane][WL] words (66 KiB x WL, L2-resident), read coalesced from a scalar base +
//     the lane's offset: the only VALU work of a byte step is the step itself - 1 DPP + 2 per word, + 1 for a word with a
//     self-loop (a per-word uniform test: large automata have few);
//   * exception edges stay SPARSE (CSR lists, as lower_nfa emits them beyond 4096 positions): only in byte steps where some
//     lane holds a live exception position (one ballot) the lanes walk theirs and OR the target bits into the wave's
//     accumulator in LDS, which the owning lanes then merge and clear - a byte costs what its live exception edges cost;
//   * the set is dense where it is populated and SKIPPED where it is not: a set that has died (every 16 bytes: one ballot)
//     stays dead until the next '\n' in line mode, so the wave looks for that '\n' 64 bytes per instruction instead of
//     stepping - the common case on text for anchored patterns.
// Same line-mode automaton as the other NFA engines: a 1 is shifted into position 0 on every byte and only the '\n' row
// contains position 0; gap positions instead of a CHAIN mask.
#include "kernels_common.hpp"

namespace rrx {
namespace dev {
namespace {

constexpr int kWaveThreads = 256;                   // four strings per workgroup

// FRONT: self-loops and exception positions only occur in word 0 of a lane (what `.*`-like prefixes and optional heads give:
// they sit at the front of the automaton) - their tests then cost 3 VALU per byte, not 2 per word.
template <int WL, bool FRONT>
struct WaveNfa {
    uint32_t fin[WL], self[WL], exc[WL];
    const uint32_t *__restrict__ rows;     // HBM/L2 [257][64][WL]: a row per byte VALUE (row 256: the line-mode '\n' row) - the
                                           //   byte is wave-uniform, so the row's address is a scalar base + the lane's offset:
                                           //   no class lookup, no address arithmetic on the VALU
    const uint32_t *__restrict__ xoff, *__restrict__ xtgt;
    uint32_t *acc;                         // LDS [64][WL]: this wave's exception accumulator (all zero between steps)
    uint32_t any_exc;                      // (a uniform word, not a bool member: the test per byte stays scalar)
    int lane;
    uint32_t lane_byte_off;                // lane * WL * 4: this lane's words inside a row

    static size_t lds_bytes(const WaveNfaDevice &) { return (size_t)(kWaveThreads / 64) * 64 * WL * 4; }
    __device__ void load(const WaveNfaDevice &p, uint8_t *lds) {
        lane = threadIdx.x & 63;
        lane_byte_off = (uint32_t)lane * WL * 4u;
        const int wave = threadIdx.x >> 6;
        uint32_t *a = reinterpret_cast<uint32_t *>(lds);
        for (int i = threadIdx.x; i < (kWaveThreads / 64) * 64 * WL; i += kWaveThreads) a[i] = 0;
        rows = p.Bbyte;
        acc = a + (size_t)wave * 64 * WL;
#pragma unroll
        for (int i = 0; i < WL; i++) {
            fin[i] = p.masks[(0 * 64 + lane) * WL + i];
            self[i] = p.masks[(1 * 64 + lane) * WL + i];
            exc[i] = p.masks[(2 * 64 + lane) * WL + i];
        }
        xoff = p.xoff; xtgt = p.xtgt; any_exc = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.exc_words);
    }
    __device__ __forceinline__ bool accepting(const uint32_t (&s)[WL]) const {
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < WL; i++) a |= s[i] & fin[i];
        return __ballot(a != 0) != 0;
    }
    __device__ __forceinline__ bool alive(const uint32_t (&s)[WL]) const {
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < WL; i++) a |= s[i];
        return __ballot(a != 0) != 0;
    }
    // The B row of a byte value (the same byte in every lane): a scalar base + this lane's 32-bit offset - global_load ... saddr
    struct Row { uint32_t b[WL]; };
    __device__ __forceinline__ Row fetch(uint32_t c) const {
        // (scalar 64-bit base + this lane's 32-bit byte offset: the form global_load takes with an SGPR base, no VALU add)
        const char *rb = reinterpret_cast<const char *>(rows) + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)c) * (256u * WL);
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(rb + lane_byte_off);
        Row r;
#pragma unroll
        for (int i = 0; i < WL; i++) r.b[i] = rp[i];
        return r;
    }
    __device__ __forceinline__ void advance(uint32_t (&s)[WL], uint32_t c) const { step(s, fetch(c)); }
    __device__ __forceinline__ void step(uint32_t (&s)[WL], const Row &row) const {
        const uint32_t (&b)[WL] = row.b;
        // lane l's lowest word takes its carry from lane l - 1's highest (DPP wave_shr:1, lane 0 gets 0).  Nothing is injected
        // into position 0: the kernels put {position 0} in place at every line start themselves (the '\n' branch exists anyway)
        const uint32_t below = __builtin_amdgcn_update_dpp(0u, s[WL - 1], 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
        uint32_t t[WL];
        t[0] = __builtin_amdgcn_alignbit(s[0], below, 31);
#pragma unroll
        for (int i = 1; i < WL; i++) t[i] = __builtin_amdgcn_alignbit(s[i], s[i - 1], 31);
        if (FRONT) t[0] |= s[0] & self[0];
        else {
#pragma unroll
            for (int i = 0; i < WL; i++) t[i] |= s[i] & self[i];
        }
        if (any_exc) {
            uint32_t e = s[0] & exc[0];
            if (!FRONT) {
#pragma unroll
                for (int i = 1; i < WL; i++) e |= s[i] & exc[i];
            }
            if (__ballot(e != 0)) {                              // some lane holds a live exception position
#pragma unroll
                for (int i = 0; i < WL; i++) {
                    uint32_t ei = s[i] & exc[i];
                    while (ei) {
                        const uint32_t p = ((uint32_t)lane * WL + i) * 32u + (uint32_t)__ffs(ei) - 1u;
                        ei &= ei - 1;
                        for (uint32_t k = xoff[p], hi = xoff[p + 1]; k < hi; k++) { const uint32_t q = xtgt[k]; atomicOr(&acc[q >> 5], 1u << (q & 31)); }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int i = 0; i < WL; i++) {
                    const uint32_t x = acc[lane * WL + i];
                    if (x) { t[i] |= x; acc[lane * WL + i] = 0; }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
#pragma unroll
        for (int i = 0; i < WL; i++) s[i] = t[i] & b[i];
    }
};

// The text of a wave: every lane wants the same byte.  16-byte aligned chunks from a wave-uniform address, the current chunk
// kept, so that ragged heads and tails (after a skipped dead line, behind the stripe) cost a select per byte, not a load.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// byte k (0..15, wave-uniform but not a constant) of a chunk: scalar selects, so that a per-byte loop need not be unrolled -
// sixteen inlined copies of a byte step with its exception path are tens of KiB of code, more than the instruction cache holds
__device__ __forceinline__ uint32_t chunk_byte(const u32x4 &v, uint32_t k) {
    const uint32_t w = k < 8 ? (k < 4 ? v.x : v.y) : (k < 12 ? v.z : v.w);
    return (w >> (8 * (k & 3))) & 0xffu;
}
struct TextFeed {
    const uint8_t *__restrict__ bytes;
    size_t limit;                          // bytes in the buffer: no chunk is read beyond it
    bool aligned;                          // the buffer starts on a dword boundary (a corpus: on a 16-byte one; explicit items may not)
    u32x4 cur = {0, 0, 0, 0};              // in SGPRs (a scalar load): the bytes, '\n' tests and row addresses derived from it are scalar too
    size_t cur_at = ~(size_t)0;
    __device__ __forceinline__ u32x4 chunk(size_t at) {           // at: wave-uniform, 16-byte aligned, at + 16 <= limit
        if (at != cur_at) {
            const uint64_t a = reinterpret_cast<uint64_t>(bytes + at);
            const uint64_t ua = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
            asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cur) : "s"(ua) : "memory");
            cur_at = at;
        }
        return cur;
    }
    __device__ __forceinline__ uint32_t at(size_t pos) {
        const size_t c = pos & ~(size_t)15;
        if (!aligned || c + 16 > limit) return (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes[pos]);      // (the buffer's last, partial chunk)
        const u32x4 v = chunk(c);
        const uint32_t k = (uint32_t)pos & 15u;
        const uint32_t w = k < 8 ? (k < 4 ? v.x : v.y) : (k < 12 ? v.z : v.w);
        return (w >> (8 * (k & 3))) & 0xffu;
    }
};

// One wave per stripe (the stripe geometry and the ownership rule are the lane kernel's, at wave granularity); every lane
// mirrors the (uniform) result bookkeeping, lane 0 writes.
template <int WL, bool FRONT>
__global__ __launch_bounds__(kWaveThreads) void match_stripes_wave_kernel(WaveNfaDevice prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                           uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                                           size_t nstripes, uint32_t *__restrict__ accept_bits) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    WaveNfa<WL, FRONT> eng;
    eng.load(prog, smem);
    __syncthreads();
    const size_t g = (size_t)blockIdx.x * (kWaveThreads / 64) + (size_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (g >= nstripes) return;                                   // whole waves leave together (no barrier below)
    const size_t start = g * (size_t)stripe;
    const size_t stripe_end = start + stripe;
    const size_t my_end = stripe_end < nbytes ? stripe_end : nbytes;
    const uint64_t my_base = stripe_base[g];
    const bool fresh = (my_base & kFreshStripe) != 0;
    uint32_t s[WL];
    auto line_start = [&]() {                                    // {position 0}: the state set at the start of a line
#pragma unroll
        for (int i = 0; i < WL; i++) s[i] = 0;
        if (eng.lane == 0) s[0] = 1u;
    };
    line_start();
    if (!fresh) s[0] = 0;                                        // inside somebody else's line: dead until the first '\n'
    Results res;
    res.begin(line_of(my_base), !fresh, accept_bits);
    res.writer = eng.lane == 0;

    auto one = [&](uint32_t c) {
        if (c == '\n') { res.push(1, eng.accepting(s) ? 1u : 0u); if (res.bits >> 30) res.flush(); line_start(); }
        else eng.advance(s, c);
    };
    TextFeed feed;
    feed.bytes = bytes; feed.limit = nbytes; feed.aligned = true;
    // a dead set stays dead until the next '\n': look for it 64 bytes per instruction.  -> true: pos is at a '\n' below `end`
    auto skip_to_newline = [&](size_t &pos, size_t end) -> bool {
        while (pos < end) {
            const size_t q = pos + (size_t)eng.lane;
            const bool nl = q < end && bytes[q] == '\n';
            const uint64_t m = __ballot(nl);
            if (m) { pos += (size_t)(__ffsll((long long)m) - 1); return true; }
            pos += 64;
        }
        return false;
    };
    size_t pos = start;
    while (pos < my_end) {
        if (!eng.alive(s)) {
            if (!skip_to_newline(pos, my_end)) break;            // no '\n' left in my stripe
            one('\n');
            pos++;
            continue;
        }
        if (!(pos & 15) && pos + 16 <= my_end) {
            const u32x4 v = feed.chunk(pos);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            // does the chunk hold a '\n'?  (scalar: the exact zero-byte test on w ^ 0x0a0a0a0a)  If not - lines are long for
            // automata of this size - its 16 steps need no test per byte
            uint32_t any_nl = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) { const uint32_t x = w[q] ^ 0x0a0a0a0au; any_nl |= ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu); }
            if (!any_nl) {
                // the rows of the next kAhead bytes are requested before the first of their (dependent) steps
                constexpr int kAhead = WL <= 4 ? 8 : WL <= 8 ? 4 : 2;
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += kAhead) {
                    typename WaveNfa<WL, FRONT>::Row rows[kAhead];
#pragma unroll
                    for (int k = 0; k < kAhead; k++) rows[k] = eng.fetch((w[(k0 + k) >> 2] >> (8 * ((k0 + k) & 3))) & 0xffu);
#pragma unroll
                    for (int k = 0; k < kAhead; k++) eng.step(s, rows[k]);
                }
            } else {
#pragma unroll 1
                for (uint32_t k = 0; k < 16; k++) one(chunk_byte(v, k));
            }
            pos += 16;
        } else {
            one(feed.at(pos));
            pos++;
        }
    }
    res.flush();

    // ---- follow my last line past the stripe end (same ownership rule as the lane kernel)
    const bool started = fresh || res.seen > 0;
    if (started && bytes[my_end - 1] != '\n') {
        bool ended = false;
        pos = my_end;
        while (pos < nbytes && !ended) {
            if (!eng.alive(s)) { ended = skip_to_newline(pos, nbytes); break; }      // dead: only the line's end matters
            const uint32_t c = feed.at(pos);
            if (c == '\n') ended = true;
            else { eng.advance(s, c); pos++; }
        }
        res.push(1, eng.accepting(s) ? 1u : 0u);                 // '\n' or the end of the corpus ends the line
    }
    res.finish();
}

// One wave per explicit item ('\n' ordinary: the plain class table, nothing shifted into position 0).
template <int WL, bool FRONT>
__global__ __launch_bounds__(kWaveThreads) void match_extents_wave_kernel(WaveNfaDevice prog, const uint8_t *__restrict__ bytes,
                                                                           const uint64_t *__restrict__ off, size_t nitems, uint32_t trim,
                                                                           uint8_t *__restrict__ accept) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    WaveNfa<WL, FRONT> eng;
    eng.load(prog, smem);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * (kWaveThreads / 64) + (size_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (i >= nitems) return;
    size_t b = off[i], e = off[i + 1];
    e = e - b >= trim ? e - trim : b;
    uint32_t s[WL];
#pragma unroll
    for (int k = 0; k < WL; k++) s[k] = 0;
    if (eng.lane == 0) s[0] = 1u;
    TextFeed feed;                                               // (no chunk beyond the item's end: the buffer's extent is not known here)
    feed.bytes = bytes; feed.limit = e; feed.aligned = (reinterpret_cast<uintptr_t>(bytes) & 3) == 0;      // (a scalar load wants a dword-aligned address)
    size_t pos = b;
    while (pos < e) {
        if (feed.aligned && !(pos & 15) && pos + 16 <= e) {
            if (!eng.alive(s)) break;                            // dead for good: nothing revives an item
            const u32x4 v = feed.chunk(pos);
#pragma unroll 1
            for (uint32_t k = 0; k < 16; k++) eng.advance(s, chunk_byte(v, k));      // 0x00 and >= 0x80: empty rows
            pos += 16;
        } else {
            eng.advance(s, feed.at(pos));
            pos++;
        }
    }
    const bool ok = eng.accepting(s);
    if (eng.lane == 0) accept[i] = ok ? 1 : 0;
}

// ============================================================================================ sparse live sets
// The reference keeps large state sets in Roaring bitmaps because they are SPARSE (README.md:18-21,57; NFA.cc:77-85 gathers
// only the rows of the live states).  The engine above is dense: a byte costs the whole set, populated or not.  This one
// keeps, per string, the LIST OF ITS NON-EMPTY BLOCKS - a bit mask `live` in a scalar register, one bit per block of 2048
// positions - and a byte costs work only for the blocks that are live, that receive a carry from the block below, or that an
// exception edge reaches:
//   * layout by blocks: block i (a "row") = positions [2048 i, 2048 i + 2048) = register s[i], lane l its bits
//     [32 l, 32 l + 32); the carry out of a row (bit 31 of lane 63) goes into lane 0 of the next row as a scalar;
//   * a skipped row costs a scalar test and a branch; a live one 7 VALU and a 256-byte row read;
//   * after its step a row stays in the list only if some lane is non-zero (one compare).
// MEASURED (profiles/r03_wave_engines.txt): the bookkeeping costs more than it saves at these sizes - 32 blocks at most, each a
// single register - also on texts where one block of four or eight is live: (a|b)*a(a|b){5000} over 30-120 byte lines
// 12.5 GB/s against 37 GB/s for the dense engine, {16000} 7.4 against 18.9.  The dense engine's own sparsity measure - a dead
// set skips to the next '\n' - is the one that pays.  Kept as an explicit engine (RRX_ENGINE_NFA_SPARSE, never AUTO's choice)
// with the full parity suite, as the list-of-blocks form SURVEY 8(f).4 describes.
//   * B rows: per byte CLASS in LDS, [class][row][lane], when they fit (48 KiB; LROWS) - a live row's read then takes LDS
//     latency, and nothing can be requested ahead for rows that are not known to be live - else per byte value in HBM/L2.
constexpr size_t kSparseRowsLdsBudget = 48 * 1024;
template <int WR, bool LROWS>
struct SparseNfa {
    uint32_t fin[WR], self[WR], exc[WR];
    const uint32_t *__restrict__ rows;     // HBM/L2 [257][WR][64] (!LROWS)
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
    lds_u32_ptr lrows;                     // LDS [ncls][WR][64] (LROWS), already advanced to this lane
    const uint8_t *cls;                    // LDS [256] (LROWS)
    const uint32_t *__restrict__ xoff, *__restrict__ xtgt;
    uint32_t *acc;                         // LDS [WR][64] + 1: this wave's exception accumulator, then the mask of the rows it hit
    bool any_exc;
    int lane;

    static size_t lds_bytes(const WaveNfaDevice &p) {
        return (LROWS ? (size_t)p.ncls * WR * 64 * 4 + 256 : 0) + (size_t)(kWaveThreads / 64) * (64 * WR + 1) * 4;
    }
    __device__ void load(const WaveNfaDevice &p, uint8_t *lds) {
        lane = threadIdx.x & 63;
        const int wave = threadIdx.x >> 6;
        uint32_t *a = reinterpret_cast<uint32_t *>(lds);
        if (LROWS) {
            const uint32_t n = p.ncls * WR * 64u;
            for (uint32_t i = threadIdx.x; i < n; i += kWaveThreads) a[i] = p.Bcls[i];
            uint8_t *c = reinterpret_cast<uint8_t *>(a + n);
            for (int i = threadIdx.x; i < 256; i += kWaveThreads) c[i] = p.cls[i];
            cls = c;
            lrows = (lds_u32_ptr)(__attribute__((address_space(3))) uint8_t *)lds + lane;
            a += n + 64;
        }
        for (int i = threadIdx.x; i < (kWaveThreads / 64) * (64 * WR + 1); i += kWaveThreads) a[i] = 0;
        rows = p.Bbyte;
        acc = a + (size_t)wave * (64 * WR + 1);
#pragma unroll
        for (int i = 0; i < WR; i++) {
            fin[i] = p.masks[(0 * WR + i) * 64 + lane];
            self[i] = p.masks[(1 * WR + i) * 64 + lane];
            exc[i] = p.masks[(2 * WR + i) * 64 + lane];
        }
        xoff = p.xoff; xtgt = p.xtgt; any_exc = p.exc_words != 0;
    }
    __device__ __forceinline__ bool accepting(const uint32_t (&s)[WR], uint32_t live) const {
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < WR; i++)
            if (live & (1u << i)) { asm volatile(""); a |= s[i] & fin[i]; }
        return __ballot(a != 0) != 0;
    }
    // c: the same byte in every lane.  live: bit i = row i may be non-zero (rows not in it ARE zero).
    __device__ __forceinline__ void advance(uint32_t (&s)[WR], uint32_t &live, uint32_t c) const {
        const uint32_t *rp = rows + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)c) * (64u * WR) + lane;
        lds_u32_ptr lp = lrows;
        if (LROWS) lp += (uint32_t)__builtin_amdgcn_readfirstlane((int)cls[c]) * (64u * WR);
        uint32_t hit = 0;                                         // rows an exception edge reaches
        if (any_exc) {
            uint32_t e = 0;
#pragma unroll
            for (int i = 0; i < WR; i++)
                if (live & (1u << i)) { asm volatile(""); e |= s[i] & exc[i]; }
            if (__ballot(e != 0)) {
#pragma unroll
                for (int i = 0; i < WR; i++) {
                    uint32_t ei = (live & (1u << i)) ? s[i] & exc[i] : 0u;
                    while (ei) {
                        const uint32_t p = ((uint32_t)i * 64u + (uint32_t)lane) * 32u + (uint32_t)__ffs(ei) - 1u;
                        ei &= ei - 1;
                        for (uint32_t k = xoff[p], hi = xoff[p + 1]; k < hi; k++) {
                            const uint32_t q = xtgt[k];
                            atomicOr(&acc[q >> 5], 1u << (q & 31));
                            atomicOr(&acc[64 * WR], 1u << (q >> 11));
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                hit = (uint32_t)__builtin_amdgcn_readfirstlane((int)acc[64 * WR]);
                if (lane == 0) acc[64 * WR] = 0;
            }
        }
        uint32_t now = 0, carry = 0;                              // carry: bit 31 of lane 63 of the row below (its OLD value)
#pragma unroll
        for (int i = 0; i < WR; i++) {
            const uint32_t bit = 1u << i;
            if (!((live | hit) & bit) && !carry) continue;        // empty, and nothing comes in: stays empty
            asm volatile("");
            const uint32_t old = (live & bit) ? s[i] : 0u;
            const uint32_t out = (live & bit) ? ((uint32_t)__builtin_amdgcn_readlane((int)old, 63) >> 31) : 0u;
            const uint32_t below = __builtin_amdgcn_update_dpp(carry ? 0x80000000u : 0u, old, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            uint32_t t = __builtin_amdgcn_alignbit(old, below, 31) | (old & self[i]);
            if (hit & bit) { const uint32_t x = acc[i * 64 + lane]; t |= x; acc[i * 64 + lane] = 0; }
            t &= LROWS ? lp[i * 64] : rp[i * 64];
            s[i] = t;
            if (__ballot(t != 0)) now |= bit;
            carry = out;
        }
        live = now;
    }
};

template <int WR, bool LROWS>
__global__ __launch_bounds__(kWaveThreads) void match_stripes_sparse_kernel(WaveNfaDevice prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                             uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                                             size_t nstripes, uint32_t *__restrict__ accept_bits) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    SparseNfa<WR, LROWS> eng;
    eng.load(prog, smem);
    __syncthreads();
    const size_t g = (size_t)blockIdx.x * (kWaveThreads / 64) + (size_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (g >= nstripes) return;
    const size_t start = g * (size_t)stripe;
    const size_t stripe_end = start + stripe;
    const size_t my_end = stripe_end < nbytes ? stripe_end : nbytes;
    const uint64_t my_base = stripe_base[g];
    const bool fresh = (my_base & kFreshStripe) != 0;
    uint32_t s[WR];
    uint32_t live;
    auto line_start = [&]() {                                    // {position 0}: one block in the list
#pragma unroll
        for (int i = 0; i < WR; i++) s[i] = 0;
        if (eng.lane == 0) s[0] = 1u;
        live = 1u;
    };
    line_start();
    if (!fresh) { s[0] = 0; live = 0; }                          // inside somebody else's line: empty until the first '\n'
    Results res;
    res.begin(line_of(my_base), !fresh, accept_bits);
    res.writer = eng.lane == 0;
    auto one = [&](uint32_t c) {
        if (c == '\n') { res.push(1, eng.accepting(s, live) ? 1u : 0u); if (res.bits >> 30) res.flush(); line_start(); }
        else eng.advance(s, live, c);
    };
    TextFeed feed;
    feed.bytes = bytes; feed.limit = nbytes; feed.aligned = true;
    auto skip_to_newline = [&](size_t &pos, size_t end) -> bool {
        while (pos < end) {
            const size_t q = pos + (size_t)eng.lane;
            const bool nl = q < end && bytes[q] == '\n';
            const uint64_t m = __ballot(nl);
            if (m) { pos += (size_t)(__ffsll((long long)m) - 1); return true; }
            pos += 64;
        }
        return false;
    };
    size_t pos = start;
    while (pos < my_end) {
        if (!live) {                                             // the list is empty: nothing to step until the next '\n'
            if (!skip_to_newline(pos, my_end)) break;
            one('\n');
            pos++;
            continue;
        }
        if (!(pos & 15) && pos + 16 <= my_end) {
            const u32x4 v = feed.chunk(pos);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 16; k++) one((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
            pos += 16;
        } else {
            one(feed.at(pos));
            pos++;
        }
    }
    res.flush();
    const bool started = fresh || res.seen > 0;
    if (started && bytes[my_end - 1] != '\n') {
        bool ended = false;
        pos = my_end;
        while (pos < nbytes && !ended) {
            if (!live) { ended = skip_to_newline(pos, nbytes); break; }
            const uint32_t c = feed.at(pos);
            if (c == '\n') ended = true;
            else { eng.advance(s, live, c); pos++; }
        }
        res.push(1, eng.accepting(s, live) ? 1u : 0u);
    }
    res.finish();
}
template <int WR, bool LROWS>
__global__ __launch_bounds__(kWaveThreads) void match_extents_sparse_kernel(WaveNfaDevice prog, const uint8_t *__restrict__ bytes,
                                                                             const uint64_t *__restrict__ off, size_t nitems, uint32_t trim,
                                                                             uint8_t *__restrict__ accept) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    SparseNfa<WR, LROWS> eng;
    eng.load(prog, smem);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * (kWaveThreads / 64) + (size_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (i >= nitems) return;
    size_t b = off[i], e = off[i + 1];
    e = e - b >= trim ? e - trim : b;
    uint32_t s[WR];
#pragma unroll
    for (int k = 0; k < WR; k++) s[k] = 0;
    if (eng.lane == 0) s[0] = 1u;
    uint32_t live = 1u;
    TextFeed feed;
    feed.bytes = bytes; feed.limit = e; feed.aligned = (reinterpret_cast<uintptr_t>(bytes) & 3) == 0;
    for (size_t pos = b; pos < e && live; pos++) eng.advance(s, live, feed.at(pos));
    const bool ok = live && eng.accepting(s, live);
    if (eng.lane == 0) accept[i] = ok ? 1 : 0;
}
template <int WR, bool LROWS>
int launch_sparse_stripes(const WaveNfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base, size_t nstripes,
                          uint32_t *accept, void *stream) {
    const size_t lds = SparseNfa<WR, LROWS>::lds_bytes(p);
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(match_stripes_sparse_kernel<WR, LROWS>), lds);
    if (e != hipSuccess) return (int)e;
    const size_t per_block = kWaveThreads / 64, blocks = (nstripes + per_block - 1) / per_block;
    hipLaunchKernelGGL((match_stripes_sparse_kernel<WR, LROWS>), dim3((unsigned)blocks), dim3(kWaveThreads), lds, (hipStream_t)stream, p, bytes, nbytes, stripe, stripe_base,
                       nstripes, accept);
    return (int)hipGetLastError();
}
template <int WR, bool LROWS>
int launch_sparse_extents(const WaveNfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept, void *stream) {
    const size_t lds = SparseNfa<WR, LROWS>::lds_bytes(p);
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(match_extents_sparse_kernel<WR, LROWS>), lds);
    if (e != hipSuccess) return (int)e;
    const size_t per_block = kWaveThreads / 64, blocks = (nitems + per_block - 1) / per_block;
    hipLaunchKernelGGL((match_extents_sparse_kernel<WR, LROWS>), dim3((unsigned)blocks), dim3(kWaveThreads), lds, (hipStream_t)stream, p, bytes, off, nitems, trim, accept);
    return (int)hipGetLastError();
}

template <int WL, bool FRONT>
int launch_wave_stripes(const WaveNfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base, size_t nstripes,
                        uint32_t *accept, void *stream) {
    const size_t lds = WaveNfa<WL, FRONT>::lds_bytes(p);
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(match_stripes_wave_kernel<WL, FRONT>), lds);
    if (e != hipSuccess) return (int)e;
    const size_t per_block = kWaveThreads / 64, blocks = (nstripes + per_block - 1) / per_block;
    hipLaunchKernelGGL((match_stripes_wave_kernel<WL, FRONT>), dim3((unsigned)blocks), dim3(kWaveThreads), lds, (hipStream_t)stream, p, bytes, nbytes, stripe, stripe_base,
                       nstripes, accept);
    return (int)hipGetLastError();
}
template <int WL, bool FRONT>
int launch_wave_extents(const WaveNfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept, void *stream) {
    const size_t lds = WaveNfa<WL, FRONT>::lds_bytes(p);
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(match_extents_wave_kernel<WL, FRONT>), lds);
    if (e != hipSuccess) return (int)e;
    const size_t per_block = kWaveThreads / 64, blocks = (nitems + per_block - 1) / per_block;
    hipLaunchKernelGGL((match_extents_wave_kernel<WL, FRONT>), dim3((unsigned)blocks), dim3(kWaveThreads), lds, (hipStream_t)stream, p, bytes, off, nitems, trim, accept);
    return (int)hipGetLastError();
}

}  // namespace

uint32_t wave_words_per_lane(uint32_t words) {                   // the instantiated width that holds `words` 32-bit words in 64 lanes
    const uint32_t need = (words + 63) / 64;
    for (uint32_t w : {1u, 2u, 3u, 4u, 6u, 8u, 12u, 16u, 24u, 32u}) if (need <= w) return w;
    return 0;
}

#define RRX_WAVE_DISPATCH(CALL)                                                                                      \
    switch (p.WL) {                                                                                                   \
    case 1: return CALL(1); case 2: return CALL(2); case 3: return CALL(3); case 4: return CALL(4); case 6: return CALL(6);       \
    case 8: return CALL(8); case 12: return CALL(12); case 16: return CALL(16); case 24: return CALL(24); case 32: return CALL(32); \
    default: return (int)hipErrorInvalidValue;                                                                        \
    }
int match_stripes_wave_nfa(const WaveNfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                           size_t nstripes, uint32_t *accept, void *stream) {
    if (!nstripes) return 0;
    const bool front = !(p.self_words & ~1u) && !(p.exc_words & ~1u);
#define CALL(W) (front ? launch_wave_stripes<W, true>(p, bytes, nbytes, stripe, stripe_base, nstripes, accept, stream) \
                       : launch_wave_stripes<W, false>(p, bytes, nbytes, stripe, stripe_base, nstripes, accept, stream))
    RRX_WAVE_DISPATCH(CALL)
#undef CALL
}
int match_extents_wave_nfa(const WaveNfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                           void *stream) {
    if (!nitems) return 0;
    const bool front = !(p.self_words & ~1u) && !(p.exc_words & ~1u);
#define CALL(W) (front ? launch_wave_extents<W, true>(p, bytes, off, nitems, trim, accept, stream) \
                       : launch_wave_extents<W, false>(p, bytes, off, nitems, trim, accept, stream))
    RRX_WAVE_DISPATCH(CALL)
#undef CALL
}
#undef RRX_WAVE_DISPATCH

// the sparse form: WR rows of 2048 positions (p.WL holds WR; p.masks / p.Bbyte are laid out by rows)
uint32_t sparse_rows(uint32_t words) {
    const uint32_t need = (words + 63) / 64;
    for (uint32_t w : {1u, 2u, 4u, 8u, 16u, 32u}) if (need <= w) return w;
    return 0;
}
#define RRX_SPARSE_DISPATCH(CALL)                                                                                     \
    switch (p.WL) {                                                                                                   \
    case 1: return CALL(1); case 2: return CALL(2); case 4: return CALL(4); case 8: return CALL(8); case 16: return CALL(16); case 32: return CALL(32); \
    default: return (int)hipErrorInvalidValue;                                                                        \
    }
int match_stripes_sparse_nfa(const WaveNfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                             size_t nstripes, uint32_t *accept, void *stream) {
    if (!nstripes) return 0;
    const bool lrows = p.Bcls && (size_t)p.ncls * p.WL * 64 * 4 <= kSparseRowsLdsBudget;
#define CALL(W) (lrows ? launch_sparse_stripes<W, true>(p, bytes, nbytes, stripe, stripe_base, nstripes, accept, stream) \
                       : launch_sparse_stripes<W, false>(p, bytes, nbytes, stripe, stripe_base, nstripes, accept, stream))
    RRX_SPARSE_DISPATCH(CALL)
#undef CALL
}
int match_extents_sparse_nfa(const WaveNfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                             void *stream) {
    if (!nitems) return 0;
    const bool lrows = p.Bcls && (size_t)p.ncls * p.WL * 64 * 4 <= kSparseRowsLdsBudget;
#define CALL(W) (lrows ? launch_sparse_extents<W, true>(p, bytes, off, nitems, trim, accept, stream) \
                       : launch_sparse_extents<W, false>(p, bytes, off, nitems, trim, accept, stream))
    RRX_SPARSE_DISPATCH(CALL)
#undef CALL
}
#undef RRX_SPARSE_DISPATCH

}  // namespace dev
}  // namespace rrx
