// kernels_coop.hip — the group-cooperative NFA engine: a state set spread over a group of 16, 32 or 64 lanes (the wave-resident
// engine for larger automata: kernels_wave.hip).
// Shared device code: kernels_common.hpp.
#include "kernels_common.hpp"

namespace rrx {
namespace dev {
namespace {

// ============================================================================================ group-cooperative NFA
// For automata too large for one lane's registers (513 ... 2048 positions): G = 16 or 32 neighbouring lanes hold ONE state
// set, lane l of the group the positions [64 l, 64 l + 64) as two 32-bit words, so a wave steps 4 or 2 strings at a time.
// (Larger automata: one string per wave, kernels_wave.hip.)  Round 3 rebuilt the step after the wave-resident engine:
//   * text: the lanes of a group load the same 16 bytes (one address per group);
//   * B rows per byte VALUE in LDS, [byte][lane of the group] 8-byte words (32 / 64 KiB): the row address is byte * row
//     bytes + the lane's offset, no class lookup in front of it;
//   * shift: the word of the lane below arrives by DPP - row_shr:1 for G = 16 (a group IS a DPP row: its lane 0 gets 0
//     for free), wave_shr:1 and one select for G = 32;
//   * nothing is injected per byte: {position 0} is put in place at every line start (the '\n' handling exists anyway),
//     and a 16-byte chunk in which no group of the wave has a '\n' - lines are long for automata of this size - is
//     stepped without any per-byte test;
//   * exception rows stay in HBM/L2 ([row][lane] words, read coalesced), one live exception position per group and turn;
//     the test for a live one is a single AND + compare when the automaton's exception positions all sit in lane 0's
//     first word (FRONT: what `.*`-like prefixes and optional heads give);
//   * verdict: ballot over the group's lanes, only in byte steps where some group of the wave sits on a '\n'.
// MODE 0: exception positions anywhere; 1 (FRONT): only in word 0 of the group's lane 0; 2 (INIT): position 0 is the ONLY one.
// Position 0 is live exactly on the first byte of a line, so in MODE 2 its exception row is ORed in on that byte alone and
// the per-byte step carries no exception test at all - no compare, no ballot, no branch in the dependent chain.
template <int G, int MODE>
struct GroupNfa {
    uint32_t fin0, fin1, self0, self1, exc0, exc1;
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
    lds_u32_ptr rows;                      // LDS [256][G][2], already advanced to this lane
    const uint16_t *__restrict__ xidx;     // HBM/L2 [nbits]: exception row of a position
    const uint2 *__restrict__ X;           // HBM/L2 [n_exc][G]
    bool any_exc;
    uint32_t x0a = 0, x0b = 0;             // MODE 2: this lane's words of position 0's exception row
    int lane, lig, gbase;                  // lane of the wave, lane of the group, the group's first lane
    uint64_t gmask;                        // the group's lanes in a ballot

    static size_t lds_bytes(const GroupNfaDevice &) { return (size_t)256 * G * 8; }
    __device__ void load(const GroupNfaDevice &p, uint8_t *lds) {
        uint4 *r = reinterpret_cast<uint4 *>(lds);
        const uint4 *src = reinterpret_cast<const uint4 *>(p.Bbyte);
        const int n = 256 * G * 8 / 16;
        for (int i = threadIdx.x; i < n; i += blockDim.x) r[i] = src[i];
        lane = threadIdx.x & 63; lig = lane & (G - 1); gbase = lane - lig;
        rows = (lds_u32_ptr)(__attribute__((address_space(3))) uint8_t *)lds + 2 * lig;
        gmask = ((1ull << G) - 1ull) << gbase;
        const uint2 *m = reinterpret_cast<const uint2 *>(p.masks);
        uint2 v;
        v = m[0 * G + lig]; fin0 = v.x; fin1 = v.y;
        v = m[1 * G + lig]; self0 = v.x; self1 = v.y;
        v = m[2 * G + lig]; exc0 = v.x; exc1 = v.y;
        xidx = p.xidx; X = reinterpret_cast<const uint2 *>(p.X); any_exc = p.n_exc != 0;
        if (MODE == 2) { const uint2 r0 = X[(size_t)xidx[0] * G + lig]; x0a = r0.x; x0b = r0.y; }
    }
    // every lane of the group must be active
    __device__ __forceinline__ bool accepting(uint32_t s0, uint32_t s1) const {
        return (__ballot(((s0 & fin0) | (s1 & fin1)) != 0) & gmask) != 0;
    }
    __device__ __forceinline__ void line_start(uint32_t &s0, uint32_t &s1) const { s0 = lig == 0 ? 1u : 0u; s1 = 0; }
    // c is the same in all lanes of a group (0x00 and >= 0x80: empty rows); first: this is the first byte of a line (MODE 2)
    __device__ __forceinline__ void advance(uint32_t &s0, uint32_t &s1, uint32_t c, bool first = false) const {
        const uint2 b = make_uint2(rows[c * (2 * G)], rows[c * (2 * G) + 1]);
        uint32_t below;
        if (G == 16) below = __builtin_amdgcn_update_dpp(0u, s1, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
        else {
            below = __builtin_amdgcn_update_dpp(0u, s1, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
            if (lig == 0) below = 0u;
        }
        uint32_t t0 = __builtin_amdgcn_alignbit(s0, below, 31) | (s0 & self0);
        uint32_t t1 = __builtin_amdgcn_alignbit(s1, s0, 31) | (s1 & self1);
        if (MODE == 2) { if (first) { t0 |= x0a; t1 |= x0b; } }
        else if (any_exc) {
            uint32_t e0 = s0 & exc0, e1 = MODE == 1 ? 0u : s1 & exc1;
            uint64_t live = __ballot((e0 | e1) != 0);            // (scalar: the lanes of the wave with a live exception position)
            while (live) {                                       // the same in every active lane
                const uint64_t mine = live & gmask;
                const int src = mine ? __ffsll((long long)mine) - 1 : lane;
                const uint32_t w0 = __shfl(e0, src, 64), w1 = __shfl(e1, src, 64);
                if (mine) {
                    const int bit = w0 ? __ffs(w0) - 1 : 32 + __ffs(w1) - 1;
                    if (lane == src) { if (bit < 32) e0 &= ~(1u << bit); else e1 &= ~(1u << (bit - 32)); }
                    const uint2 row = X[(size_t)xidx[(src - gbase) * 64 + bit] * G + lig];
                    t0 |= row.x; t1 |= row.y;
                }
                live = __ballot((e0 | e1) != 0);
            }
        }
        s0 = t0 & b.x; s1 = t1 & b.y;
    }
};

// One group per stripe (the stripe geometry and the result path are the lane kernel's, at group granularity); every
// lane of a group mirrors the result bookkeeping, its lane 0 alone writes.
template <int G, int MODE>
__global__ __launch_bounds__(256) void match_stripes_group_kernel(GroupNfaDevice prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                   uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                                   uint32_t *__restrict__ accept_bits) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    GroupNfa<G, MODE> eng;
    eng.load(prog, smem);
    __syncthreads();
    const size_t g = ((size_t)blockIdx.x * 256 + threadIdx.x) / G;
    const size_t start = g * (size_t)stripe;
    if (start >= nbytes) return;                               // whole groups leave together
    const size_t stripe_end = start + stripe;
    const size_t my_end = stripe_end < nbytes ? stripe_end : nbytes;
    const uint64_t my_base = stripe_base[g];
    const bool fresh = (my_base & kFreshStripe) != 0;
    uint32_t s0, s1;
    eng.line_start(s0, s1);
    bool first = fresh;                                        // the next byte is the first of a line (the same in all lanes of a group)
    if (!fresh) s0 = 0;                                        // inside somebody else's line: dead until the first '\n'
    Results res;
    res.begin(line_of(my_base), !fresh, accept_bits);
    res.writer = eng.lig == 0;

    auto one = [&](uint32_t c) {
        const bool isnl = c == '\n';
        uint32_t a0 = s0, a1 = s1;
        eng.advance(a0, a1, c, first);
        first = false;
        if (__ballot(isnl)) {                                  // some group of the wave ends a line on this byte
            const bool a = eng.accepting(s0, s1);
            if (isnl) { res.push(1, a ? 1u : 0u); if (res.bits >> 30) res.flush(); eng.line_start(a0, a1); first = true; }
        }
        s0 = a0; s1 = a1;
    };
    // ---- the lines inside my stripe: stripes are multiples of 16 bytes, only the corpus end leaves a tail
    size_t pos = start;
    for (; pos + 16 <= my_end; pos += 16) {
        const uint4 v = *reinterpret_cast<const uint4 *>(bytes + pos);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t nl = 0;                                       // exact zero-byte test on w ^ 0x0a0a0a0a
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t x = w[q] ^ 0x0a0a0a0au; nl |= ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu); }
        if (!__ballot(nl != 0 || (MODE == 2 && first))) {      // no group of the wave meets a '\n' in its 16 bytes (or stands at a line start)
#pragma unroll
            for (int k = 0; k < 16; k++) eng.advance(s0, s1, (w[k >> 2] >> (8 * (k & 3))) & 0xffu);
        } else {
#pragma unroll 1
            for (int k = 0; k < 16; k++) {
                const uint32_t wk = k < 8 ? (k < 4 ? v.x : v.y) : (k < 12 ? v.z : v.w);
                one((wk >> (8 * (k & 3))) & 0xffu);
            }
        }
    }
    for (; pos < my_end; pos++) one(bytes[pos]);
    res.flush();

    // ---- follow my last line past the stripe end (same ownership rule as the lane kernel)
    const bool started = fresh || res.seen > 0;
    if (started && bytes[my_end - 1] != '\n') {
        bool ended = false;
        // 16 bytes per load (pos is 16-byte aligned here unless the corpus ended inside my stripe): one byte per load was a
        // chain of memory round trips as long as half a line - 350 bytes on the 500-900 byte lines these automata are for
        while (!ended && !(pos & 15) && pos + 16 <= nbytes) {
            const uint4 v = *reinterpret_cast<const uint4 *>(bytes + pos);
#pragma unroll 1
            for (int k = 0; k < 16 && !ended; k++) {
                const uint32_t wk = k < 8 ? (k < 4 ? v.x : v.y) : (k < 12 ? v.z : v.w);
                const uint32_t c = (wk >> (8 * (k & 3))) & 0xffu;
                if (c == '\n') ended = true;
                else { eng.advance(s0, s1, c, first); first = false; }
            }
            pos += 16;
        }
        for (; pos < nbytes && !ended; pos++) {
            const uint32_t c = bytes[pos];
            if (c == '\n') ended = true;
            else { eng.advance(s0, s1, c, first); first = false; }
        }
        res.push(1, eng.accepting(s0, s1) ? 1u : 0u);          // '\n' or the end of the corpus ends the line
    }
    res.finish();
}

// One group per explicit item ('\n' an ordinary byte).
template <int G, int MODE>
__global__ __launch_bounds__(256) void match_extents_group_kernel(GroupNfaDevice prog, const uint8_t *__restrict__ bytes,
                                                                   const uint64_t *__restrict__ off, size_t nitems, uint32_t trim,
                                                                   uint8_t *__restrict__ accept) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    GroupNfa<G, MODE> eng;
    eng.load(prog, smem);
    __syncthreads();
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (i >= nitems) return;
    size_t b = off[i], e = off[i + 1];
    e = e - b >= trim ? e - trim : b;
    uint32_t s0, s1;
    eng.line_start(s0, s1);
    for (size_t pos = b; pos < e; pos++) eng.advance(s0, s1, bytes[pos], pos == b);
    const bool ok = eng.accepting(s0, s1);
    if (eng.lig == 0) accept[i] = ok ? 1 : 0;
}

}  // namespace

template <int G, int MODE>
static int launch_group_stripes(const GroupNfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                                size_t nstripes, uint32_t *accept, void *stream) {
    const size_t lds = GroupNfa<G, MODE>::lds_bytes(p);
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(match_stripes_group_kernel<G, MODE>), lds);
    if (e != hipSuccess) return (int)e;
    const size_t per_block = 256 / G, blocks = (nstripes + per_block - 1) / per_block;
    hipLaunchKernelGGL((match_stripes_group_kernel<G, MODE>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p, bytes, nbytes, stripe, stripe_base, accept);
    return (int)hipGetLastError();
}
template <int G, int MODE>
static int launch_group_extents(const GroupNfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                                void *stream) {
    const size_t lds = GroupNfa<G, MODE>::lds_bytes(p);
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(match_extents_group_kernel<G, MODE>), lds);
    if (e != hipSuccess) return (int)e;
    const size_t per_block = 256 / G, blocks = (nitems + per_block - 1) / per_block;
    hipLaunchKernelGGL((match_extents_group_kernel<G, MODE>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p, bytes, off, nitems, trim, accept);
    return (int)hipGetLastError();
}
#define RRX_GROUP_DISPATCH(FN, ...)                                                                      \
    switch (p.G * 4 + p.exc_mode) {                                                                      \
    case 16 * 4 + 0: return FN<16, 0>(__VA_ARGS__); case 16 * 4 + 1: return FN<16, 1>(__VA_ARGS__); case 16 * 4 + 2: return FN<16, 2>(__VA_ARGS__); \
    case 32 * 4 + 0: return FN<32, 0>(__VA_ARGS__); case 32 * 4 + 1: return FN<32, 1>(__VA_ARGS__); case 32 * 4 + 2: return FN<32, 2>(__VA_ARGS__); \
    default: return (int)hipErrorInvalidValue;                                                           \
    }
int match_stripes_group_nfa(const GroupNfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                            size_t nstripes, uint32_t *accept, void *stream) {
    if (!nstripes) return 0;
    RRX_GROUP_DISPATCH(launch_group_stripes, p, bytes, nbytes, stripe, stripe_base, nstripes, accept, stream)
}
int match_extents_group_nfa(const GroupNfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                            void *stream) {
    if (!nitems) return 0;
    RRX_GROUP_DISPATCH(launch_group_extents, p, bytes, off, nitems, trim, accept, stream)
}
#undef RRX_GROUP_DISPATCH
// ---- the NFA lane engines are built in four parts by width (kernels_nfa.inc); the entry points pick the part
#define RRX_NFA_PARTS(name, ARGS_DECL, ARGS)                                                        \
    int name##_part0 ARGS_DECL; int name##_part1 ARGS_DECL; int name##_part2 ARGS_DECL; int name##_part3 ARGS_DECL; \
    int name ARGS_DECL { return p.W <= 2 ? name##_part0 ARGS : p.W <= 4 ? name##_part1 ARGS : p.W <= 8 ? name##_part2 ARGS : name##_part3 ARGS; }
RRX_NFA_PARTS(match_stripes_nfa,
              (const NfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base, size_t nstripes, uint32_t *accept, void *stream),
              (p, bytes, nbytes, stripe, stripe_base, nstripes, accept, stream))
RRX_NFA_PARTS(match_onepass_nfa,
              (const NfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, size_t nstripes, uint32_t *counts, uint32_t *slabs, void *stream),
              (p, bytes, nbytes, stripe, nstripes, counts, slabs, stream))
RRX_NFA_PARTS(match_extents_nfa,
              (const NfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept, void *stream),
              (p, bytes, off, nitems, trim, accept, stream))
RRX_NFA_PARTS(recheck_escaped_nfa,
              (const NfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base, size_t nstripes, const uint32_t *escaped_bits,
               size_t nlines, const uint64_t *list, const unsigned long long *escaped_total, size_t cap, uint32_t *accept_bits, void *stream),
              (p, bytes, nbytes, stripe, stripe_base, nstripes, escaped_bits, nlines, list, escaped_total, cap, accept_bits, stream))
RRX_NFA_PARTS(match_long_nfa,
              (const NfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t chunk, uint32_t nchunks, void *scratch, uint8_t *accept, void *stream),
              (p, bytes, nbytes, chunk, nchunks, scratch, accept, stream))
#undef RRX_NFA_PARTS
// chunking of one long string for the NFA form: at most 65536 chunks of 1 KiB or more, and at most 256 MiB of rows
// (chunks x positions x words x 4 bytes); the composition levels ping-pong between the rows and a second area of half
// their size (rounded up to a whole relation)
size_t long_nfa_scratch_bytes(const NfaDevice &p, size_t nbytes, uint32_t *chunk, uint32_t *nchunks) {
    uint32_t c = 1024;
    const size_t per_chunk = (size_t)p.nbits * p.W * 4;
    while ((nbytes + c - 1) / c > 65536 || ((nbytes + c - 1) / c) * per_chunk > ((size_t)256 << 20)) c <<= 1;
    *chunk = c;
    *nchunks = (uint32_t)((nbytes + c - 1) / c);
    return ((size_t)*nchunks + ((size_t)*nchunks + 1) / 2) * per_chunk;
}
}  // namespace dev
}  // namespace rrx
