// kernels_coop.hip — the group-cooperative NFA engine: a state set spread over a group of 8, 16 or 32 lanes (the wave-resident
// engine for larger automata: kernels_wave.hip).
// Shared device code: kernels_common.hpp.
#include "kernels_common.hpp"

namespace rrx {
namespace dev {
namespace {

// ============================================================================================ group-cooperative NFA
// For automata too large for one lane's registers (513 ... 8192 positions): G = 8, 16 or 32 neighbouring lanes hold ONE state
// set, lane l of the group the words [K l, K l + K) of it (K = 2 ... 8), so a wave steps 8, 4 or 2 strings at a time.
// (Larger automata: one string per wave, kernels_wave.hip.)  Round 3 rebuilt the step after the wave-resident engine; round 4
// made the words per lane a parameter (K was 2: 604 positions took 16 lanes at 59 % use - now 8 lanes x 3 words; 5003 took a
// whole wave - now 32 lanes x 5 words, two strings per wave) and moved the B rows from byte values to byte classes:
//   * text: the lanes of a group load the same 16 bytes (one address per group);
//   * B rows per byte CLASS in LDS, [class][lane of the group][K] words, behind a 256-byte class map (K words per lane and 256
//     byte values would not fit); the rows do not depend on the state: their reads run ahead of the step;
//   * slots (word index within a lane) that no lane needs a mask for are stepped without it: a slot whose B rows are all ones
//     for every class >= 1 takes no AND and no LDS read (what `.{n}` and `(a|b){n}` tails are made of) - a byte of class 0 then
//     cannot clear it, so it marks the line DEAD instead (one flag per lane, looked at with the verdict); a slot without self
//     loops takes no self term;
//   * shift: slot k takes the top bit of slot k - 1, slot 0 that of the lane below by DPP - row_shr:1 (a group of 16 IS a DPP
//     row: its lane 0 gets 0 for free; groups of 8: one select), wave_shr:1 and one select for G = 32;
//   * nothing is injected per byte: {position 0} is put in place at every line start (the '\n' handling exists anyway),
//     and a 16-byte chunk in which no group of the wave has a '\n' - lines are long for automata of this size - is
//     stepped without any per-byte test;
//   * exception rows stay in HBM/L2 ([row][lane][K] words, read coalesced), one live exception position per group and turn;
//   * verdict: ballot over the group's lanes, only in byte steps where some group of the wave sits on a '\n'.
// MODE 0: exception positions anywhere; 2 (INIT): position 0 is the ONLY one.  Position 0 is live exactly on the first byte of
// a line, so in MODE 2 its exception row is ORed in on that byte alone and the per-byte step carries no exception test at all -
// no compare, no ballot, no branch in the dependent chain.
template <int G, int K, int MODE, bool SLOT0>
struct GroupNfa {
    uint32_t fin[K], self[K], exc[K], x0[K];
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
    typedef const __attribute__((address_space(3))) uint8_t *lds_u8_ptr;
    lds_u32_ptr rows;                      // LDS [ncls][G][K], already advanced to this lane
    lds_u8_ptr cls;                        // LDS [256]
    const uint16_t *__restrict__ xidx;     // HBM/L2 [nbits]: exception row of a position
    const uint32_t *__restrict__ X;        // HBM/L2 [n_exc][G][K]
    uint32_t self_slots, b_slots, exc_slots;
    // SLOT0: slot 0 alone carries B rows and self loops (GroupNfaDevice::b_slots / self_slots say so): the other slots are stepped
    // by the shift alone, and a byte of class 0 kills the line by flag (`dead`) since it cannot clear them.
    int lane, lig, gbase;                  // lane of the wave, lane of the group, the group's first lane
    uint64_t gmask;                        // the group's lanes in a ballot

    static size_t lds_bytes(const GroupNfaDevice &p) { return (size_t)p.ncls * G * K * 4 + 256; }
    __device__ void load(const GroupNfaDevice &p, uint8_t *lds) {
        uint32_t *r = reinterpret_cast<uint32_t *>(lds);
        const int n = (int)(p.ncls * G * K);
        for (int i = threadIdx.x; i < n; i += blockDim.x) r[i] = p.Bcls[i];
        uint8_t *c = lds + (size_t)n * 4;
        for (int i = threadIdx.x; i < 256; i += blockDim.x) c[i] = p.cls[i];
        lane = threadIdx.x & 63; lig = lane & (G - 1); gbase = lane - lig;
        rows = (lds_u32_ptr)(__attribute__((address_space(3))) uint8_t *)lds + K * lig;
        cls = (lds_u8_ptr)(__attribute__((address_space(3))) uint8_t *)c;
        gmask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << gbase;
        self_slots = p.self_slots; b_slots = p.b_slots; exc_slots = p.exc_slots;
#pragma unroll
        for (int k = 0; k < K; k++) {
            fin[k] = p.masks[(0 * G + lig) * K + k]; self[k] = p.masks[(1 * G + lig) * K + k]; exc[k] = p.masks[(2 * G + lig) * K + k];
            x0[k] = 0;
        }
        xidx = p.xidx; X = p.X;
        if (MODE == 2) {
#pragma unroll
            for (int k = 0; k < K; k++) x0[k] = X[((size_t)xidx[0] * G + lig) * K + k];
        }
    }
    struct State { uint32_t s[K]; bool dead; };
    // every lane of the group must be active
    __device__ __forceinline__ bool accepting(const State &st) const {
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < K; k++) any |= st.s[k] & fin[k];
        return !st.dead && (__ballot(any != 0) & gmask) != 0;
    }
    __device__ __forceinline__ void line_start(State &st) const {
#pragma unroll
        for (int k = 0; k < K; k++) st.s[k] = 0;
        if (lig == 0) st.s[0] = 1u;
        st.dead = false;
    }
    __device__ __forceinline__ void kill(State &st) const {
#pragma unroll
        for (int k = 0; k < K; k++) st.s[k] = 0;
    }
    // The part of a byte step that does not depend on the state: the byte's class, then its B words (SLOT0: slot 0's only).  The
    // kernels request these for several bytes ahead of the dependent steps.  c is the same in all lanes of a group.
    struct Row { uint32_t b[SLOT0 ? 1 : K]; bool kills; };
    __device__ __forceinline__ uint32_t class_of(uint32_t c) const { return cls[c]; }          // 0x00 and >= 0x80: class 0
    __device__ __forceinline__ Row fetch(uint32_t cl) const {
        lds_u32_ptr row = rows + cl * (uint32_t)(G * K);
        Row r;
#pragma unroll
        for (int k = 0; k < (SLOT0 ? 1 : K); k++) r.b[k] = row[k];
        r.kills = cl == 0u;
        return r;
    }
    __device__ __forceinline__ void advance(State &st, uint32_t c, bool first = false) const { step(st, fetch(class_of(c)), first); }
    // first: this is the first byte of a line (MODE 2)
    __device__ __forceinline__ void step(State &st, const Row &r, bool first = false) const {
        if (SLOT0) st.dead = st.dead || r.kills;
        uint32_t below;
        if (G == 16) below = __builtin_amdgcn_update_dpp(0u, st.s[K - 1], 0x111 /* row_shr:1 */, 0xf, 0xf, true);
        else if (G == 8) {
            below = __builtin_amdgcn_update_dpp(0u, st.s[K - 1], 0x111 /* row_shr:1 */, 0xf, 0xf, true);
            if (lig == 0) below = 0u;
        } else {
            below = __builtin_amdgcn_update_dpp(0u, st.s[K - 1], 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
            if (lig == 0) below = 0u;
        }
        uint32_t t[K];
        t[0] = __builtin_amdgcn_alignbit(st.s[0], below, 31);
#pragma unroll
        for (int k = 1; k < K; k++) t[k] = __builtin_amdgcn_alignbit(st.s[k], st.s[k - 1], 31);
#pragma unroll
        for (int k = 0; k < (SLOT0 ? 1 : K); k++) t[k] |= st.s[k] & self[k];
        if (MODE == 2) {
            if (first) {
#pragma unroll
                for (int k = 0; k < K; k++) t[k] |= x0[k];
            }
        } else if (exc_slots) {
            // my lowest live exception position as a position of the set, or none
            uint32_t mine_pos = 0xffffffffu;
            uint32_t e[K];
#pragma unroll
            for (int k = K - 1; k >= 0; k--) {
                e[k] = st.s[k] & exc[k];
                if (e[k]) mine_pos = (uint32_t)((lig * K + k) * 32) + (uint32_t)__ffs((int)e[k]) - 1u;
            }
            uint64_t live = __ballot(mine_pos != 0xffffffffu);   // (scalar: the lanes of the wave with a live exception position)
            while (live) {                                       // the same in every active lane
                const uint64_t mine = live & gmask;
                const int src = mine ? __ffsll((long long)mine) - 1 : lane;
                const uint32_t pos = __shfl(mine_pos, src, 64);
                if (mine) {
                    if (lane == src) {                           // that position is done: my next one
                        const uint32_t slot = (pos >> 5) - (uint32_t)(lig * K), bit = pos & 31u;
                        mine_pos = 0xffffffffu;
#pragma unroll
                        for (int k = K - 1; k >= 0; k--) {
                            if ((uint32_t)k == slot) e[k] &= ~(1u << bit);
                            if (e[k]) mine_pos = (uint32_t)((lig * K + k) * 32) + (uint32_t)__ffs((int)e[k]) - 1u;
                        }
                    }
                    const uint32_t *xr = X + ((size_t)xidx[pos] * G + lig) * K;
#pragma unroll
                    for (int k = 0; k < K; k++) t[k] |= xr[k];
                }
                live = __ballot(mine_pos != 0xffffffffu);
            }
        }
#pragma unroll
        for (int k = 0; k < K; k++) st.s[k] = (SLOT0 && k > 0) ? t[k] : (t[k] & r.b[SLOT0 ? 0 : k]);
    }
};

// One group per SPAN of consecutive stripes (the stripe geometry and the result path are the lane kernel's, at group granularity:
// the index knows the line number at every stripe start, so a group may begin at any of them and run through several); every
// lane of a group mirrors the result bookkeeping, its lane 0 alone writes.  The corpus' stripe is chosen for engines that give a
// LANE a stripe (a quarter of a million of them at least): a group of 32 lanes per 2 KiB of 700-byte lines walks a sixth of its
// text a second time behind its stripe end - span makes the group's share 8-16 KiB while enough groups remain to fill the chip
// ((a|b)*a(a|b){5000}, 256 MiB: 49.9 -> 76 GB/s).
template <int G, int K, int MODE, bool SLOT0>
__global__ __launch_bounds__(256) void match_stripes_group_kernel(GroupNfaDevice prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                   uint32_t stripe1, uint32_t span, const uint64_t *__restrict__ stripe_base,
                                                                   uint32_t *__restrict__ accept_bits) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef GroupNfa<G, K, MODE, SLOT0> Eng;
    Eng eng;
    eng.load(prog, smem);
    __syncthreads();
    const size_t g = ((size_t)blockIdx.x * 256 + threadIdx.x) / G;
    const size_t stripe = (size_t)stripe1 * span;              // my share of the text: `span` stripes of the index
    const size_t start = g * stripe;
    if (start >= nbytes) return;                               // whole groups leave together
    const size_t stripe_end = start + stripe;
    const size_t my_end = stripe_end < nbytes ? stripe_end : nbytes;
    const uint64_t my_base = stripe_base[g * span];
    const bool fresh = (my_base & kFreshStripe) != 0;
    typename Eng::State st;
    eng.line_start(st);
    bool first = fresh;                                        // the next byte is the first of a line (the same in all lanes of a group)
    if (!fresh) eng.kill(st);                                  // inside somebody else's line: dead until the first '\n'
    Results res;
    res.begin(line_of(my_base), !fresh, accept_bits);
    res.writer = eng.lig == 0;

    auto one_row = [&](uint32_t c, const typename Eng::Row &row) {
        const bool isnl = c == '\n';
        typename Eng::State a = st;
        eng.step(a, row, first);
        first = false;
        if (__ballot(isnl)) {                                  // some group of the wave ends a line on this byte
            const bool ok = eng.accepting(st);
            if (isnl) { res.push(1, ok ? 1u : 0u); if (res.bits >> 30) res.flush(); eng.line_start(a); first = true; }
        }
        st = a;
    };
    auto one = [&](uint32_t c) { one_row(c, eng.fetch(eng.class_of(c))); };
    // ---- the lines inside my stripe: stripes are multiples of 16 bytes, only the corpus end leaves a tail
    size_t pos = start;
    for (; pos + 16 <= my_end; pos += 16) {
        const uint4 v = *reinterpret_cast<const uint4 *>(bytes + pos);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t nl = 0;                                       // exact zero-byte test on w ^ 0x0a0a0a0a
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t x = w[q] ^ 0x0a0a0a0au; nl |= ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu); }
        if (!__ballot(nl != 0 || (MODE == 2 && first))) {      // no group of the wave meets a '\n' in its 16 bytes (or stands at a line start)
            // the classes of all sixteen bytes, then four bytes' B words at a time, ahead of the dependent steps
            uint32_t cl[16];
#pragma unroll
            for (int k = 0; k < 16; k++) cl[k] = eng.class_of((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const typename Eng::Row r0 = eng.fetch(cl[4 * q]), r1 = eng.fetch(cl[4 * q + 1]), r2 = eng.fetch(cl[4 * q + 2]), r3 = eng.fetch(cl[4 * q + 3]);
                eng.step(st, r0); eng.step(st, r1); eng.step(st, r2); eng.step(st, r3);
            }
        } else {
            // some line ends in these sixteen bytes (or begins with them): a text word at a time, its four classes and rows ahead
            // of the four steps (a byte at a time left each step behind two LDS round trips: a twentieth of the chunks, a quarter
            // of the time on 700-byte lines)
#pragma unroll 1
            for (int q = 0; q < 4; q++) {
                const uint32_t wq = q < 2 ? (q == 0 ? v.x : v.y) : (q == 2 ? v.z : v.w);
                const uint32_t c0 = wq & 0xffu, c1 = (wq >> 8) & 0xffu, c2 = (wq >> 16) & 0xffu, c3 = wq >> 24;
                const uint32_t l0 = eng.class_of(c0), l1 = eng.class_of(c1), l2 = eng.class_of(c2), l3 = eng.class_of(c3);
                const typename Eng::Row r0 = eng.fetch(l0), r1 = eng.fetch(l1), r2 = eng.fetch(l2), r3 = eng.fetch(l3);
                one_row(c0, r0); one_row(c1, r1); one_row(c2, r2); one_row(c3, r3);
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
                else { eng.advance(st, c, first); first = false; }
            }
            pos += 16;
        }
        for (; pos < nbytes && !ended; pos++) {
            const uint32_t c = bytes[pos];
            if (c == '\n') ended = true;
            else { eng.advance(st, c, first); first = false; }
        }
        res.push(1, eng.accepting(st) ? 1u : 0u);              // '\n' or the end of the corpus ends the line
    }
    res.finish();
}

// One group per explicit item ('\n' an ordinary byte).
template <int G, int K, int MODE, bool SLOT0>
__global__ __launch_bounds__(256) void match_extents_group_kernel(GroupNfaDevice prog, const uint8_t *__restrict__ bytes,
                                                                   const uint64_t *__restrict__ off, size_t nitems, uint32_t trim,
                                                                   uint8_t *__restrict__ accept) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef GroupNfa<G, K, MODE, SLOT0> Eng;
    Eng eng;
    eng.load(prog, smem);
    __syncthreads();
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (i >= nitems) return;
    size_t b = off[i], e = off[i + 1];
    e = e - b >= trim ? e - trim : b;
    typename Eng::State st;
    eng.line_start(st);
    for (size_t pos = b; pos < e; pos++) eng.advance(st, bytes[pos], pos == b);
    const bool ok = eng.accepting(st);
    if (eng.lig == 0) accept[i] = ok ? 1 : 0;
}

}  // namespace

template <int G, int K, int MODE, bool SLOT0>
static int launch_group_stripes(const GroupNfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                                size_t nstripes, uint32_t *accept, void *stream) {
    const size_t lds = GroupNfa<G, K, MODE, SLOT0>::lds_bytes(p);
    if (lds > 128 * 1024) return (int)hipErrorInvalidValue;
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(match_stripes_group_kernel<G, K, MODE, SLOT0>), lds);
    if (e != hipSuccess) return (int)e;
    // stripes per group: up to 16 KiB of text, while twice as many groups remain as the chip holds at a time (256 CUs x 28 waves)
    const size_t resident_groups = (size_t)256 * 28 * (64 / G);
    uint32_t span = 1;
    while ((size_t)stripe * span * 2 <= 16384 && nstripes / (2 * (size_t)span) >= 2 * resident_groups) span *= 2;
    const size_t ngroups = (nstripes + span - 1) / span;
    const size_t per_block = 256 / G, blocks = (ngroups + per_block - 1) / per_block;
    hipLaunchKernelGGL((match_stripes_group_kernel<G, K, MODE, SLOT0>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p, bytes, nbytes, stripe, span,
                       stripe_base, accept);
    return (int)hipGetLastError();
}
template <int G, int K, int MODE, bool SLOT0>
static int launch_group_extents(const GroupNfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                                void *stream) {
    const size_t lds = GroupNfa<G, K, MODE, SLOT0>::lds_bytes(p);
    if (lds > 128 * 1024) return (int)hipErrorInvalidValue;
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(match_extents_group_kernel<G, K, MODE, SLOT0>), lds);
    if (e != hipSuccess) return (int)e;
    const size_t per_block = 256 / G, blocks = (nitems + per_block - 1) / per_block;
    hipLaunchKernelGGL((match_extents_group_kernel<G, K, MODE, SLOT0>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p, bytes, off, nitems, trim, accept);
    return (int)hipGetLastError();
}
// (G, K) as device.hpp: group_geometry hands them out; three builds of each: exceptions anywhere, position 0 the only exception,
// and that with slot 0 the only slot that carries masks
#define RRX_GROUP_FORM(FN, GG, KK, ...)                                                                  \
    if (p.G == GG && p.K == KK)                                                                          \
        return p.exc_mode != 2 ? FN<GG, KK, 0, false>(__VA_ARGS__) : slot0 ? FN<GG, KK, 2, true>(__VA_ARGS__) : FN<GG, KK, 2, false>(__VA_ARGS__);
#define RRX_GROUP_DISPATCH(FN, ...)                                                                      \
    const bool slot0 = ((p.b_slots | p.self_slots) & ~1u) == 0u;                                         \
    RRX_GROUP_FORM(FN, 8, 3, __VA_ARGS__) RRX_GROUP_FORM(FN, 8, 4, __VA_ARGS__)                                             \
    RRX_GROUP_FORM(FN, 16, 3, __VA_ARGS__) RRX_GROUP_FORM(FN, 16, 4, __VA_ARGS__)                                           \
    RRX_GROUP_FORM(FN, 32, 3, __VA_ARGS__) RRX_GROUP_FORM(FN, 32, 4, __VA_ARGS__) RRX_GROUP_FORM(FN, 32, 5, __VA_ARGS__)    \
    RRX_GROUP_FORM(FN, 32, 8, __VA_ARGS__)                                                                                  \
    return (int)hipErrorInvalidValue;
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
#undef RRX_GROUP_FORM
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
