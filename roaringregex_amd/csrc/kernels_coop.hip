// kernels_coop.hip — the group-cooperative NFA engine: a state set spread over a group of 16, 32 or 64 lanes (the wave-resident
// engine for larger automata: kernels_wave.hip).
// Shared device code: kernels_common.hpp.
#include "kernels_common.hpp"

namespace rrx {
namespace dev {
namespace {

// ============================================================================================ group-cooperative NFA
// For automata too large for one lane's registers (513 ... 4096 positions): G = 16, 32 or 64 neighbouring lanes hold ONE
// state set, lane l of the group the positions [64 l, 64 l + 64) as two 32-bit words, so a wave steps 4, 2 or 1 strings
// at a time (the first version gave every string a whole wave whatever its automaton's size and read its rows from L2:
// tens of MB/s).  The same line-mode automaton as the lane engine: a 1 is shifted into position 0 on every byte and only
// the '\n' row contains position 0; gap positions instead of a CHAIN mask.
//   * text: the lanes of a group load the same 16 bytes (one address per group);
//   * B rows: per byte CLASS (the '\n' row last), [class][lane of the group] 8-byte words in LDS: consecutive lanes read
//     consecutive words;
//   * shift: the word of the lane below arrives by DPP wave_shr:1 (no LDS traffic), the group's lane 0 gets the injected 1;
//   * exception rows stay in HBM/L2 ([row][lane] words, read coalesced), one live exception position per group and turn:
//     the loop runs while ANY group of the wave has one left, groups without one idle through it;
//   * verdict: ballot over the group's lanes, only in byte steps where some group of the wave sits on a '\n'.
template <int G>
struct GroupNfa {
    uint32_t fin0, fin1, self0, self1, exc0, exc1;
    const uint2 *rows;                     // LDS [ncls][G]
    const uint8_t *cls;                    // LDS [256]
    const uint16_t *__restrict__ xidx;     // HBM/L2 [nbits]: exception row of a position
    const uint2 *__restrict__ X;           // HBM/L2 [n_exc][G]
    bool any_exc;
    int lane, lig, gbase;                  // lane of the wave, lane of the group, the group's first lane
    uint64_t gmask;                        // the group's lanes in a ballot

    static size_t lds_bytes(const GroupNfaDevice &p) { return (size_t)p.ncls * G * 8 + 256; }
    __device__ void load(const GroupNfaDevice &p, uint8_t *lds, bool line_mode) {
        uint32_t *r = reinterpret_cast<uint32_t *>(lds);
        const int n = (int)(p.ncls * G * 2);
        for (int i = threadIdx.x; i < n; i += blockDim.x) r[i] = p.Bcls[i];
        uint8_t *c = lds + (size_t)n * 4;
        const uint8_t *src = line_mode ? p.cls_line : p.cls_plain;
        for (int i = threadIdx.x; i < 256; i += blockDim.x) c[i] = src[i];
        rows = reinterpret_cast<const uint2 *>(lds); cls = c;
        lane = threadIdx.x & 63; lig = lane & (G - 1); gbase = lane - lig;
        gmask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << gbase;
        const uint2 *m = reinterpret_cast<const uint2 *>(p.masks);
        uint2 v;
        v = m[0 * G + lig]; fin0 = v.x; fin1 = v.y;
        v = m[1 * G + lig]; self0 = v.x; self1 = v.y;
        v = m[2 * G + lig]; exc0 = v.x; exc1 = v.y;
        xidx = p.xidx; X = reinterpret_cast<const uint2 *>(p.X); any_exc = p.n_exc != 0;
    }
    // every lane of the group must be active
    __device__ __forceinline__ bool accepting(uint32_t s0, uint32_t s1) const {
        return (__ballot(((s0 & fin0) | (s1 & fin1)) != 0) & gmask) != 0;
    }
    // c is the same in all lanes of a group
    template <bool LINE>
    __device__ __forceinline__ void advance(uint32_t &s0, uint32_t &s1, uint32_t c) const {
        const uint2 b = rows[(uint32_t)cls[c] * G + lig];
        uint32_t below = __builtin_amdgcn_update_dpp(0u, s1, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        if (lig == 0) below = LINE ? 0x80000000u : 0u;          // line mode: the 1 shifted into position 0
        uint32_t t0 = __builtin_amdgcn_alignbit(s0, below, 31) | (s0 & self0);
        uint32_t t1 = __builtin_amdgcn_alignbit(s1, s0, 31) | (s1 & self1);
        if (any_exc) {
            uint32_t e0 = s0 & exc0, e1 = s1 & exc1;
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
template <int G>
__global__ __launch_bounds__(256) void match_stripes_group_kernel(GroupNfaDevice prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                   uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                                   uint32_t *__restrict__ accept_bits) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    GroupNfa<G> eng;
    eng.load(prog, smem, true);
    __syncthreads();
    const size_t g = ((size_t)blockIdx.x * 256 + threadIdx.x) / G;
    const size_t start = g * (size_t)stripe;
    if (start >= nbytes) return;                               // whole groups leave together
    const size_t stripe_end = start + stripe;
    const size_t my_end = stripe_end < nbytes ? stripe_end : nbytes;
    const uint64_t my_base = stripe_base[g];
    const bool fresh = (my_base & kFreshStripe) != 0;
    uint32_t s0 = (fresh && eng.lig == 0) ? 1u : 0u, s1 = 0;   // not fresh: dead until the first '\n'
    Results res;
    res.begin(line_of(my_base), !fresh, accept_bits);
    res.writer = eng.lig == 0;

    auto one = [&](uint32_t c) {
        const bool isnl = c == '\n';
        if (__ballot(isnl)) {                                  // some group of the wave ends a line on this byte
            const bool a = eng.accepting(s0, s1);
            if (isnl) { res.push(1, a ? 1u : 0u); if (res.bits >> 30) res.flush(); }
        }
        eng.template advance<true>(s0, s1, c);
    };
    // ---- the lines inside my stripe: stripes are multiples of 16 bytes, only the corpus end leaves a tail
    size_t pos = start;
    for (; pos + 16 <= my_end; pos += 16) {
        const uint4 v = *reinterpret_cast<const uint4 *>(bytes + pos);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; k++) one((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
    }
    for (; pos < my_end; pos++) one(bytes[pos]);
    res.flush();

    // ---- follow my last line past the stripe end (same ownership rule as the lane kernel)
    const bool started = fresh || res.seen > 0;
    if (started && bytes[my_end - 1] != '\n') {
        bool ended = false;
        for (; pos < nbytes && !ended; pos++) {
            const uint32_t c = bytes[pos];
            if (c == '\n') ended = true;
            else eng.template advance<true>(s0, s1, c);
        }
        res.push(1, eng.accepting(s0, s1) ? 1u : 0u);          // '\n' or the end of the corpus ends the line
    }
    res.finish();
}

// One group per explicit item ('\n' ordinary: the plain class table, nothing shifted into position 0).
template <int G>
__global__ __launch_bounds__(256) void match_extents_group_kernel(GroupNfaDevice prog, const uint8_t *__restrict__ bytes,
                                                                   const uint64_t *__restrict__ off, size_t nitems, uint32_t trim,
                                                                   uint8_t *__restrict__ accept) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    GroupNfa<G> eng;
    eng.load(prog, smem, false);
    __syncthreads();
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (i >= nitems) return;
    size_t b = off[i], e = off[i + 1];
    e = e - b >= trim ? e - trim : b;
    uint32_t s0 = eng.lig == 0 ? 1u : 0u, s1 = 0;
    for (size_t pos = b; pos < e; pos++) eng.template advance<false>(s0, s1, bytes[pos]);      // 0x00 and >= 0x80: empty rows
    const bool ok = eng.accepting(s0, s1);
    if (eng.lig == 0) accept[i] = ok ? 1 : 0;
}

}  // namespace

template <int G>
static int launch_group_stripes(const GroupNfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                                size_t nstripes, uint32_t *accept, void *stream) {
    const size_t lds = GroupNfa<G>::lds_bytes(p);
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(match_stripes_group_kernel<G>), lds);
    if (e != hipSuccess) return (int)e;
    const size_t per_block = 256 / G, blocks = (nstripes + per_block - 1) / per_block;
    hipLaunchKernelGGL(match_stripes_group_kernel<G>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p, bytes, nbytes, stripe, stripe_base, accept);
    return (int)hipGetLastError();
}
template <int G>
static int launch_group_extents(const GroupNfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                                void *stream) {
    const size_t lds = GroupNfa<G>::lds_bytes(p);
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(match_extents_group_kernel<G>), lds);
    if (e != hipSuccess) return (int)e;
    const size_t per_block = 256 / G, blocks = (nitems + per_block - 1) / per_block;
    hipLaunchKernelGGL(match_extents_group_kernel<G>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p, bytes, off, nitems, trim, accept);
    return (int)hipGetLastError();
}
int match_stripes_group_nfa(const GroupNfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                            size_t nstripes, uint32_t *accept, void *stream) {
    if (!nstripes) return 0;
    if (GroupNfa<64>::lds_bytes(p) > kGroupLdsBudget) return (int)hipErrorInvalidValue;
    switch (p.G) {
    case 16: return launch_group_stripes<16>(p, bytes, nbytes, stripe, stripe_base, nstripes, accept, stream);
    case 32: return launch_group_stripes<32>(p, bytes, nbytes, stripe, stripe_base, nstripes, accept, stream);
    case 64: return launch_group_stripes<64>(p, bytes, nbytes, stripe, stripe_base, nstripes, accept, stream);
    default: return (int)hipErrorInvalidValue;
    }
}
int match_extents_group_nfa(const GroupNfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                            void *stream) {
    if (!nitems) return 0;
    switch (p.G) {
    case 16: return launch_group_extents<16>(p, bytes, off, nitems, trim, accept, stream);
    case 32: return launch_group_extents<32>(p, bytes, off, nitems, trim, accept, stream);
    case 64: return launch_group_extents<64>(p, bytes, off, nitems, trim, accept, stream);
    default: return (int)hipErrorInvalidValue;
    }
}
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
