// kernels_table.hip — table engines (byte-stride and stride-2), the line index, the stream compaction of the one-shot
// entry, search and the one-long-string kernels.  Shared device code: kernels_common.hpp.
#include "kernels_common.hpp"

namespace rrx {
namespace dev {
namespace {

// ============================================================================================ engines
// Line-mode engines expose
//     void load(program, lds)               cooperative table copy into LDS
//     State fresh() / State skipping()      start of a line / inside a line owned by somebody else
//     void step(State&, c, nl, acc)         consume one byte; nl = 1 iff it was '\n', acc = verdict of the
//                                           line it ended (valid when nl)

// ---- wide / classed table DFA: '\n' handling folded into the table -------------------------------
template <bool WIDE, bool CLAMP>
struct LineDfaEngine {
    static constexpr bool kStaged = true;      // results go through the workgroup's LDS window (ResultsT<true>)
    static constexpr bool kEightWaves = false;
    static constexpr int kRoundBytes = kRound;
    // Table entry: bits 0..15 = byte offset of the next row, byte 2 = 1 iff the consumed byte was '\n',
    // byte 3 = verdict of the line it ended.  (16-bit entries read with ds_read_u16 measured 3-4 % slower.)
    struct State { uint32_t e; };
    const uint8_t *tab;                    // LDS, byte-addressed
    const uint8_t *cls;                    // LDS [256] (classed form)
    uint32_t start_off, dead_off;
    uint32_t col_shift;                    // log2(bytes between neighbouring columns) = 2 + log2(copies)

    static size_t lds_bytes(const LineDfaDevice &p) { return (size_t)p.nrows * p.stride * 4 + (WIDE ? 0 : 256); }
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
    __device__ void load(const LineDfaDevice &p, uint8_t *lds) {
        uint32_t *t = reinterpret_cast<uint32_t *>(lds);
        const int n = (int)(p.nrows * p.stride);
        // In the SDWA form the low half of an entry is the ABSOLUTE LDS address of the next row, so that
        // e.word[0] + 4*c is the address to read, with no base to add per byte.
        const uint32_t base = (WIDE && !CLAMP) ? (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds : 0u;
        copy_table_to_lds(t, p.table, (uint32_t)n * 4u, base);
        if (!WIDE) {
            uint8_t *c = lds + (size_t)n * 4;
            for (int i = threadIdx.x; i < 256; i += blockDim.x) c[i] = p.cls[i];
            cls = c;
        }
        tab = lds;
        // R interleaved copies (wide form): lane l lives in copy l % R, whose dwords sit in banks = l (mod R)
        const uint32_t copy = WIDE ? (threadIdx.x & ((1u << p.rep_log2) - 1u)) * 4u : 0u;
        col_shift = 2u + (WIDE ? p.rep_log2 : 0u);
        start_off = p.start_off + base + copy;
        dead_off = base + copy;
    }
    __device__ __forceinline__ State fresh() const { return State{start_off}; }
    __device__ __forceinline__ State skipping() const { return State{dead_off}; }   // dead row: waits for '\n'
    // Byte K of text word w, fused with the result accumulation bits = (bits << nl) | acc.  A wave64 integer
    // VALU op costs 4 cycles on a CDNA4 SIMD, so the step is written as 4 VALU + 1 LDS per byte with the
    // field extractions folded into SDWA operand selects (hipcc emits 6-7 for the plain C form below):
    //     c4   = w.byte[K] << 2                 v_lshlrev_b32_sdwa   src1_sel:BYTE_K
    //     addr = e.word[0] + c4                 v_add_u32_sdwa       src0_sel:WORD_0
    //     e    = LDS[addr]                      ds_read_b32
    //     bits = bits << e.byte[2]              v_lshlrev_b32_sdwa   src0_sel:BYTE_2
    //     bits = bits |  e.byte[3]              v_or_b32_sdwa        src0_sel:BYTE_3
    template <int K>
    __device__ __forceinline__ void consume(State &st, uint32_t w, uint32_t &bits) const {
        if constexpr (WIDE && !CLAMP) {
            // One asm block per byte (separate statements made hipcc pad every byte with an s_nop).  The block
            // waits for its own LDS read; the only other memory traffic of the wave are global loads (vmcnt).
            uint32_t t0, t1;
#define RRX_STEP(SEL)                                                                                                        \
            asm volatile("v_lshlrev_b32_sdwa %[c4], %[two], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" SEL "\n\t" \
                         "v_add_u32_sdwa %[ad], %[e], %[c4] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"       \
                         "ds_read_b32 %[e], %[ad]\n\t"                                                                                  \
                         "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
                         "v_lshlrev_b32_sdwa %[b], %[e], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n\t"     \
                         "v_or_b32_sdwa %[b], %[e], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD"              \
                         : [e] "+v"(st.e), [b] "+v"(bits), [c4] "=&v"(t0), [ad] "=&v"(t1)                                               \
                         : [w] "v"(w), [two] "v"(col_shift)                                                                             \
                         : "memory")
            if constexpr (K == 0) RRX_STEP("BYTE_0");
            if constexpr (K == 1) RRX_STEP("BYTE_1");
            if constexpr (K == 2) RRX_STEP("BYTE_2");
            if constexpr (K == 3) RRX_STEP("BYTE_3");
#undef RRX_STEP
        } else {
            uint32_t nl, acc;
            step(st, (w >> (8 * K)) & 0xffu, nl, acc);
            bits = (bits << nl) | acc;
        }
    }
    __device__ __forceinline__ void consume_word(State &st, uint32_t w, uint32_t &bits) const {
        consume<0>(st, w, bits); consume<1>(st, w, bits); consume<2>(st, w, bits); consume<3>(st, w, bits);
    }
    __device__ __forceinline__ void step(State &st, uint32_t c, uint32_t &nl, uint32_t &acc) const {
        uint32_t col;
        if (WIDE) col = CLAMP ? (c < 128u ? c : 128u) : c;      // !CLAMP: the corpus holds no byte >= 0x80
        else col = cls[c];
        const uint32_t off = (st.e & 0xffffu) + (col << col_shift);      // (absolute LDS address in the SDWA form)
        st.e = (WIDE && !CLAMP) ? *reinterpret_cast<lds_u32_ptr>(off) : *reinterpret_cast<const uint32_t *>(tab + off);
        nl = (st.e >> 16) & 0xffu;
        acc = st.e >> 24;
    }
};

// ---- table DFA whose table stays in global memory (L2-resident): any automaton up to 65535 interned sets ----
struct LineDfaGlobalEngine {
    static constexpr bool kStaged = true;
    static constexpr bool kEightWaves = false;
    static constexpr int kRoundBytes = kRound;
    struct State { uint32_t e; };          // low 24 bits = index of the current row's first entry
    const uint32_t *__restrict__ tab;      // HBM / L2
    const uint8_t *cls;                    // LDS [256]
    uint32_t start_off;

    static size_t lds_bytes(const LineDfaDevice &) { return 256; }
    __device__ void load(const LineDfaDevice &p, uint8_t *lds) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) lds[i] = p.cls[i];
        cls = lds; tab = p.table; start_off = p.start_off;
    }
    __device__ __forceinline__ State fresh() const { return State{start_off}; }
    __device__ __forceinline__ State skipping() const { return State{0}; }
    __device__ __forceinline__ void step(State &st, uint32_t c, uint32_t &nl, uint32_t &acc) const {
        st.e = tab[(st.e & 0xffffffu) + cls[c]];
        nl = (st.e >> 30) & 1u;
        acc = st.e >> 31;
    }
    template <int K>
    __device__ __forceinline__ void consume(State &st, uint32_t w, uint32_t &bits) const {
        uint32_t nl, acc;
        step(st, (w >> (8 * K)) & 0xffu, nl, acc);
        bits = (bits << nl) | acc;
    }
    __device__ __forceinline__ void consume_word(State &st, uint32_t w, uint32_t &bits) const {
        consume<0>(st, w, bits); consume<1>(st, w, bits); consume<2>(st, w, bits); consume<3>(st, w, bits);
    }
};

struct PlainDfaEngine {
    struct State { uint32_t s; };
    const uint8_t *cls;     // LDS [256]
    const uint16_t *next;   // LDS [nstates][ncls]
    const uint8_t *acc;     // LDS [nstates]
    uint32_t ncls, start;

    static size_t lds_bytes(const DfaDevice &p) {
        size_t t = ((size_t)p.nstates * p.ncls * 2 + 15) & ~(size_t)15;
        return t + 256 + ((p.nstates + 15) & ~15u);
    }
    __device__ void load(const DfaDevice &p, uint8_t *lds) {
        size_t tb = ((size_t)p.nstates * p.ncls * 2 + 15) & ~(size_t)15;
        uint16_t *n = reinterpret_cast<uint16_t *>(lds);
        uint8_t *c = lds + tb;
        uint8_t *a = c + 256;
        for (int i = threadIdx.x; i < (int)(p.nstates * p.ncls); i += blockDim.x) n[i] = p.next[i];
        for (int i = threadIdx.x; i < 256; i += blockDim.x) c[i] = p.cls[i];
        for (int i = threadIdx.x; i < (int)p.nstates; i += blockDim.x) a[i] = p.acc[i];
        next = n; cls = c; acc = a; ncls = p.ncls; start = p.start;
    }
    __device__ __forceinline__ void reset(State &st) const { st.s = start; }
    __device__ __forceinline__ void kill(State &st) const { st.s = 0; }
    __device__ __forceinline__ bool accepting(const State &st) const { return acc[st.s] != 0; }
    __device__ __forceinline__ void step(State &st, uint32_t c) const { st.s = next[st.s * ncls + cls[c]]; }
};

// The same automaton with its table left in HBM/L2 (tables beyond the LDS budget: the batch kernel's "global" form,
// here for explicit items and single strings).  One dependent L2 read per byte.
struct PlainDfaGlobalEngine {
    struct State { uint32_t s; };
    const uint8_t *cls;                   // LDS [256]
    const uint16_t *__restrict__ next;    // HBM / L2 [nstates][ncls]
    const uint8_t *__restrict__ acc;      // HBM / L2 [nstates]
    uint32_t ncls, start;

    static size_t lds_bytes(const DfaDevice &) { return 256; }
    __device__ void load(const DfaDevice &p, uint8_t *lds) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) lds[i] = p.cls[i];
        cls = lds; next = p.next; acc = p.acc; ncls = p.ncls; start = p.start;
    }
    __device__ __forceinline__ void reset(State &st) const { st.s = start; }
    __device__ __forceinline__ void kill(State &st) const { st.s = 0; }
    __device__ __forceinline__ bool accepting(const State &st) const { return acc[st.s] != 0; }
    __device__ __forceinline__ void step(State &st, uint32_t c) const { st.s = next[(size_t)st.s * ncls + cls[c]]; }
};

// ============================================================================================ stride-2 table kernel
// The per-byte table step is bounded by the latency of its dependent LDS round trip (add -> ds_read -> wait, ~210
// cycles at 8 chains per SIMD).  Here ONE dependent lookup consumes TWO bytes: the pair's column comes from the
// state-independent table P (its read does not wait for the state), then e = T2[row(e)][column].  U2: 46 distinct
// pair columns of 289 class pairs, T2 = 16 KiB.  Per pair: 6 VALU + 2 LDS reads (3 VALU per byte).
struct Dfa2 {
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
    struct State { uint32_t e; };          // low 16 bits = LDS address of the current row (of this lane's copy)
    const uint16_t *P;                     // LDS (a static array at a link-time address: no base to add per pair)
    uint32_t start_off, dead_off;

    __host__ __device__ static size_t lds_bytes(const Dfa2Device &p) { return (size_t)p.nrows * p.stride * 4; }     // dynamic part: T2
    __device__ void load(const Dfa2Device &p, uint16_t *p_lds, uint8_t *t_lds, uint32_t p_bytes = kDfa2PBytes) {
        const uint32_t tbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)t_lds;
        copy_table_to_lds(p_lds, p.P, p_bytes);
        copy_table_to_lds(t_lds, p.T2, p.nrows * p.stride * 4, tbase);
        const uint32_t copy = (threadIdx.x & ((1u << p.rep_log2) - 1u)) * 4u;
        P = p_lds;
        start_off = p.start_off + tbase + copy;
        dead_off = tbase + copy;
    }
    __device__ __forceinline__ State fresh() const { return State{start_off}; }
    __device__ __forceinline__ State skipping() const { return State{dead_off}; }
    // generic pair step (tails and the walk past the stripe end)
    __device__ __forceinline__ void step2(State &st, uint32_t c1, uint32_t c2, uint32_t &lines, uint32_t &verdicts) const {
        const uint32_t col = P[c1 * kDfa2PStride + c2];
        st.e = *reinterpret_cast<lds_u32_ptr>((st.e & 0xffffu) + col);
        lines = (st.e >> 16) & 0xffu;
        verdicts = st.e >> 24;
    }
    // the four bytes of text word w (two pairs), fused with bits = (bits << lines) | verdicts.  Per pair:
    //     t    = (2 c1) * 130                v_mul_u32_u24_sdwa   src0_sel:BYTE_even
    //     idx  = t + 2 c2                    v_add_u32_sdwa       src1_sel:BYTE_odd        (byte offset into P)
    //     col  = P[idx]                      ds_read_u16                                   (does not wait for the state)
    //     addr = e.word[0] + col             v_add_u32_sdwa       src0_sel:WORD_0
    //     e    = LDS[addr]                   ds_read_b32
    //     bits = (bits << e.byte[2]) | e.byte[3]                 2 x SDWA
    __device__ __forceinline__ void consume_dword(State &st, uint32_t w, uint32_t &bits) const {
        const uint32_t w2 = w << 1;                      // every byte < 0x80: doubling stays inside the byte
        const uint32_t stride = kDfa2PStride;
        uint32_t ta, ia, tb, ib;
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(ta) : "v"(w2), "v"(stride));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(ia) : "v"(ta), "v"(w2));
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(tb) : "v"(w2), "v"(stride));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(ib) : "v"(tb), "v"(w2));
#define RRX_LDS_U16(x) (*reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(P) + (x)))
#define RRX_LDS_U32(x) (*reinterpret_cast<lds_u32_ptr>(x))
        const uint32_t ca = RRX_LDS_U16(ia);
        const uint32_t cb = RRX_LDS_U16(ib);
        uint32_t addr;
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(addr) : "v"(st.e), "v"(ca));
        st.e = RRX_LDS_U32(addr);
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(bits) : "v"(st.e), "v"(bits));
        asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(bits) : "v"(st.e), "v"(bits));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(addr) : "v"(st.e), "v"(cb));
        st.e = RRX_LDS_U32(addr);
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(bits) : "v"(st.e), "v"(bits));
        asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(bits) : "v"(st.e), "v"(bits));
    }
    // The items form (codes 0 ... 128, 128 = END OF ITEM): a byte can no longer be doubled inside the text word, so the entry index
    // c1 * 130 + c2 is made first and doubled afterwards - one VALU more per pair.
    __device__ __forceinline__ void consume_dword_items(State &st, uint32_t w, uint32_t &bits) const {
        const uint32_t stride = kDfa2PStride;
        uint32_t ta, ia, tb, ib;
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(ta) : "v"(w), "v"(stride));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(ia) : "v"(ta), "v"(w));
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(tb) : "v"(w), "v"(stride));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(ib) : "v"(tb), "v"(w));
        const uint32_t ca = RRX_LDS_U16(ia << 1);
        const uint32_t cb = RRX_LDS_U16(ib << 1);
        uint32_t addr;
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(addr) : "v"(st.e), "v"(ca));
        st.e = RRX_LDS_U32(addr);
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(bits) : "v"(st.e), "v"(bits));
        asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(bits) : "v"(st.e), "v"(bits));
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(addr) : "v"(st.e), "v"(cb));
        st.e = RRX_LDS_U32(addr);
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(bits) : "v"(st.e), "v"(bits));
        asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(bits) : "v"(st.e), "v"(bits));
    }
};

// Same stripe geometry, feed and result path as match_stripes_kernel; pairs are aligned to even byte positions
// (stripes are even-sized), a line end may fall on either byte of a pair.
// ONEPASS (rrx_match_device): no line index yet.  Every lane starts in the start state (a lane that begins inside a line
// produces a verdict for the fragment, which the compaction drops), counts its '\n' on the side, and keeps its verdict
// stream in the workgroup's slab (LocalResults).  Bytes >= 0x80 cannot index the pair table: a text word that holds one is
// rewritten with 0x00 in their place (which rejects the line just the same) under a wave-uniform branch.
// `phase` is an instrumentation hook: the shipping kernels pass NoPhaseHook (nothing is emitted); tools/probe/stamps builds
// a kernel around this body whose hook writes a timestamp per workgroup and phase (where a launch's fixed cost goes).
struct NoPhaseHook { __device__ __forceinline__ void operator()(int) const {} __device__ __forceinline__ void round(int) const {} };
enum { kPhaseEntry = 0, kPhaseTablesLoaded, kPhaseFirstRound, kPhaseMainDone, kPhaseFollowDone, kPhaseWindowOut, kPhases };
// A round's text as ONE burst of eight loads.  ASM: issued from one inline-asm block whose destinations are early-clobber, so the
// address can never share registers with a destination (the unit kernel's register allocation put the address into the last
// load's destination: a wait state inside the burst, whose loads then stop merging into one request per 128-byte line).  The
// compiler does not count loads it cannot see, so the block ends with the wait itself: nothing was ever scheduled between a
// round's request and its first use anyway (the other waves of the SIMD cover the fetch).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool ASM>
__device__ __forceinline__ void feed_load(uint4 (&buf)[kRound / 16], const uint4 *p) {
    static_assert(kRound / 16 == 8, "eight slots");
    if constexpr (!ASM) {
#pragma unroll
        for (int i = 0; i < 8; i++) buf[i] = load_text(p + i);
    } else {
        u32x4 a, b, c, d, e, f, g, h;
        asm volatile("global_load_dwordx4 %0, %8, off\n\t"
                     "global_load_dwordx4 %1, %8, off offset:16\n\t"
                     "global_load_dwordx4 %2, %8, off offset:32\n\t"
                     "global_load_dwordx4 %3, %8, off offset:48\n\t"
                     "global_load_dwordx4 %4, %8, off offset:64\n\t"
                     "global_load_dwordx4 %5, %8, off offset:80\n\t"
                     "global_load_dwordx4 %6, %8, off offset:96\n\t"
                     "global_load_dwordx4 %7, %8, off offset:112\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h)
                     : "v"(p)
                     : "memory");
        const u32x4 t[8] = {a, b, c, d, e, f, g, h};
#pragma unroll
        for (int i = 0; i < 8; i++) buf[i] = make_uint4(t[i].x, t[i].y, t[i].z, t[i].w);
    }
}
// One stripe of one lane: lane-local state only (what the caller keeps across stripes is the engine, the window and `g`).
// KB: result bits per line - 1 (accepted), or 2 (accepted, ESCAPED: the sampled-table engine, whose table does not know every
// transition; the table's line ends then shift two bits in, and everything that counts results counts bits).
template <bool ONEPASS, class PhaseHook, bool FEED_ASM = false, int KB = 1>
__device__ __forceinline__ void dfa2_stripe(const Dfa2 &eng, const size_t g, const uint64_t window_word, uint32_t *const stage, const uint32_t stage_words,
                                            const uint8_t *__restrict__ bytes, const size_t nbytes, const uint32_t stripe,
                                            const uint64_t *__restrict__ stripe_base, uint32_t *__restrict__ accept_bits,
                                            uint32_t *__restrict__ counts, uint32_t *__restrict__ slabs, const uint32_t slab_row, PhaseHook &phase,
                                            const uint32_t flush_mask = 31u) {
    const size_t start = g * (size_t)stripe;
    if (start < nbytes) {                                            // (no early return: the write-out below is collective)
    const size_t stripe_end = start + stripe;
    const size_t my_end = stripe_end < nbytes ? stripe_end : nbytes;
    bool fresh = true;
    typename std::conditional<ONEPASS, LocalResults, ResultsT<true>>::type res;
    if constexpr (ONEPASS) {
        res.begin(slabs + g, slab_row);
    } else {
        const uint64_t my_base = stripe_base[g];
        fresh = (my_base & kFreshStripe) != 0;
        res.begin_staged(line_of(my_base) * KB, window_word, !fresh, accept_bits, stage);
        res.stage_words = stage_words;
        res.drop_mask = (1u << KB) - 1u;
    }
    Dfa2::State st = fresh ? eng.fresh() : eng.skipping();
    auto clean = [](uint32_t w) -> uint32_t {                       // ONEPASS: bytes >= 0x80 -> 0x00
        if (ONEPASS && __builtin_amdgcn_ballot_w64((w & 0x80808080u) != 0)) {
            const uint32_t hi = (w & 0x80808080u) >> 7;             // 1 in every byte to clear
            w &= ~(hi * 0xffu);
        }
        return w;
    };

    size_t pos = start;
    const uint4 *src = reinterpret_cast<const uint4 *>(bytes + start);
    constexpr int kSlots = kRound / 16;
    const int rounds = (int)((my_end - start) / kRound);
    uint4 buf[kSlots];
    if (rounds > 0) feed_load<FEED_ASM>(buf, src);
    // The line that straddles my stripe end is followed into the next stripe's text (below).  Its first 128 bytes are
    // requested while the last round is still being stepped: loaded on demand, 16 bytes at a time, they were a chain of
    // L2 round trips at the end of every wave's life, and the waves of a workgroup - of the whole chip, launched together
    // and fed at the same rate - reach that point at the same time.
    uint32_t last_word = 0;                                       // the last text word of my stripe (whole rounds only)
    bool ahead = false;                                           // buf holds the 128 bytes behind my stripe
    for (int r = 0; r < rounds; r++) {
        if (r == 1) phase(kPhaseFirstRound);
        phase.round(r);
#pragma unroll
        for (int i = 0; i < kSlots; i++) {
            if (ONEPASS && __builtin_expect(__builtin_amdgcn_ballot_w64(((buf[i].x | buf[i].y | buf[i].z | buf[i].w) & 0x80808080u) != 0) != 0, 0)) {
                // some lane of the wave holds a byte >= 0x80 in this slot (one test per 16 bytes; rare on text)
                eng.consume_dword(st, clean(buf[i].x), res.bits);
                eng.consume_dword(st, clean(buf[i].y), res.bits);
                eng.consume_dword(st, clean(buf[i].z), res.bits);
                eng.consume_dword(st, clean(buf[i].w), res.bits);
            } else {
                eng.consume_dword(st, buf[i].x, res.bits);
                eng.consume_dword(st, buf[i].y, res.bits);
                if (KB == 2 && (res.bits >> 15)) res.flush();    // (two bits per line end: eight bytes can bring sixteen)
                eng.consume_dword(st, buf[i].z, res.bits);
                eng.consume_dword(st, buf[i].w, res.bits);
            }
            // All lanes flush TOGETHER every (flush_mask + 1) slots - a period the host picks from the corpus' mean line length so
            // that about eight results gather in it (short lines: every other slot; 512 bytes for long ones).  The overflow
            // check behind it is for the lanes that meet far more: left to it alone, lanes overflow at different times and the
            // wave walks the flush path at nearly every slot (5-byte lines: +1.25 VALU per byte).
            if ((((uint32_t)r * kSlots + (uint32_t)i) & flush_mask) == flush_mask) res.flush();
            else if (res.bits >> 15) res.flush();            // <= 16 more results fit before the next check
        }
        if (r + 1 < rounds) {
            feed_load<FEED_ASM>(buf, src + (r + 1) * kSlots);
        } else {
            last_word = buf[kSlots - 1].w;
            if (start + (size_t)(rounds + 1) * kRound <= nbytes) {
                feed_load<FEED_ASM>(buf, src + (r + 1) * kSlots);
                ahead = true;
            }
        }
    }
    pos += (size_t)rounds * kRound;
    phase(kPhaseMainDone);
    auto byte_at = [&](size_t q) -> uint32_t { const uint32_t b = bytes[q]; return (ONEPASS && b >= 0x80u) ? 0u : b; };

    // ---- tail of the corpus inside my stripe (only the last stripe has one): whole pairs, then an odd last byte.
    // The odd byte is paired with a virtual '\n': if it is a '\n' itself the pair reports two line ends, of which
    // only the first exists; otherwise the virtual '\n' is the end of data ending the last line, and the walk
    // below must not end it again.
    bool closed_by_end_of_data = false;
    for (; pos + 2 <= my_end; pos += 2) {
        uint32_t lines, verdicts;
        eng.step2(st, byte_at(pos), byte_at(pos + 1), lines, verdicts);
        res.bits = (res.bits << lines) | verdicts;
        if (res.bits >> (31 - 2 * KB)) res.flush();
    }
    if (pos < my_end) {
        const uint32_t b = byte_at(pos);
        uint32_t lines, verdicts;
        eng.step2(st, b, '\n', lines, verdicts);
        if (b == '\n') res.push(KB, verdicts >> KB);
        else { res.push(KB, verdicts); closed_by_end_of_data = true; }
        pos++;
    }
    res.flush();
    const uint32_t newlines = res.seen - (closed_by_end_of_data ? 1u : 0u);      // real '\n' inside my stripe

    // ---- follow my last line past the stripe end (same ownership rule as the byte kernel), pair by pair
    if (ONEPASS && res.seen == 0) fresh = g == 0 || bytes[start - 1] == '\n';   // a stripe without any '\n': whose line is it?
    const bool started = fresh || res.seen > 0;
    bool followed = false;
    const bool whole_rounds = rounds > 0 && start + (size_t)rounds * kRound == my_end;
    const uint32_t last_byte = whole_rounds ? last_word >> 24 : (uint32_t)bytes[my_end - 1];
    if (!closed_by_end_of_data && started && last_byte != '\n') {
        uint32_t lines = 0, verdicts = 0;
        if (ahead && pos == start + (size_t)(rounds + 1) * kRound - kRound) {      // (pos == my_end: the requested bytes are the next ones)
    #pragma unroll
            for (int i = 0; i < kSlots; i++) {
                if (!lines) {
                    const uint32_t w[4] = {clean(buf[i].x), clean(buf[i].y), clean(buf[i].z), clean(buf[i].w)};
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        if (!lines) eng.step2(st, (w[k >> 1] >> (16 * (k & 1))) & 0xffu, (w[k >> 1] >> (16 * (k & 1) + 8)) & 0xffu, lines, verdicts);
                    pos += 16;
                }
            }
        }
        while (pos + 16 <= nbytes && !lines) {
            const uint4 v = *reinterpret_cast<const uint4 *>(bytes + pos);
            const uint32_t w[4] = {clean(v.x), clean(v.y), clean(v.z), clean(v.w)};
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (!lines) eng.step2(st, (w[k >> 1] >> (16 * (k & 1))) & 0xffu, (w[k >> 1] >> (16 * (k & 1) + 8)) & 0xffu, lines, verdicts);
            pos += 16;
        }
        for (; pos + 2 <= nbytes && !lines; pos += 2) eng.step2(st, byte_at(pos), byte_at(pos + 1), lines, verdicts);
        if (!lines) eng.step2(st, pos < nbytes ? byte_at(pos) : '\n', '\n', lines, verdicts);   // end of data ends the line
        res.push(KB, lines == 2 * KB ? verdicts >> KB : verdicts);     // only the first line end of the pair is mine
        followed = true;
    }
    res.finish();
    phase(kPhaseFollowDone);
    if (ONEPASS)
        counts[g] = newlines | ((followed || closed_by_end_of_data) ? kExtraResult : 0u) | (last_byte == '\n' ? kEndsOnNewline : 0u);
    }
}

template <bool ONEPASS, class PhaseHook = NoPhaseHook, int KB = 1>
__device__ __forceinline__ void dfa2_body(const Dfa2Device &prog, const uint8_t *__restrict__ bytes, size_t nbytes, uint32_t stripe,
                                          const uint64_t *__restrict__ stripe_base, uint32_t *__restrict__ accept_bits,
                                          uint32_t *__restrict__ counts, uint32_t *__restrict__ slabs, PhaseHook phase = PhaseHook(),
                                          uint32_t flush_mask = 31u) {
    phase(kPhaseEntry);
    // T2 first: its entries hold 16-bit LDS addresses; the result window takes what T2 leaves of its region (16 KiB and
    // more for tables up to 30 KiB, 4 KiB at least).  The arrays are static, so P's base is a link-time constant.
    __shared__ __attribute__((aligned(16))) struct {
        uint8_t t2_and_stage[kDfa2RegionBytes];
        uint16_t p[kDfa2PBytes / 2];
    } lds;
    Dfa2 eng;
    eng.load(prog, lds.p, lds.t2_and_stage);
    const uint32_t stage_off = (uint32_t)((Dfa2::lds_bytes(prog) + 15) & ~(size_t)15);
    uint32_t *const stage = reinterpret_cast<uint32_t *>(lds.t2_and_stage + stage_off);
    const uint32_t stage_words = (kDfa2RegionBytes - stage_off) / 4;
    if (!ONEPASS)
        for (uint32_t i = threadIdx.x; i < stage_words; i += kThreads) stage[i] = 0;
    __syncthreads();
    phase(kPhaseTablesLoaded);

    const size_t g0 = (size_t)blockIdx.x * kThreads;
    uint64_t window_word = 0;
    if (!ONEPASS) window_word = (line_of(stripe_base[g0]) * KB) >> 5;       // the workgroup's first stripe exists: uniform load
    dfa2_stripe<ONEPASS, PhaseHook, false, KB>(eng, g0 + threadIdx.x, window_word, stage, stage_words, bytes, nbytes, stripe, stripe_base, accept_bits, counts, slabs,
                         gridDim.x * kThreads, phase, flush_mask);
    if (!ONEPASS) {
        // ---- write the window out: consecutive lanes, consecutive words (the atomics merge into whole lines in L2;
        // the first and the last word of the window are shared with the neighbouring workgroups)
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < stage_words; i += kThreads) {
            const uint32_t v = stage[i];
            if (v) atomicOr(&accept_bits[window_word + i], v);
        }
    }
    phase(kPhaseWindowOut);
}
// The same stripes handed out in UNITS inside the workgroup.  A launch of one generation - 1 GiB: 512 workgroups,
// two per CU - ends in a drain: a SIMD issues from its oldest wave first, its eight waves finish their stripes at 122 ... 228 us and
// for the last third of the kernel a CU runs on a few waves (profiles/r03_phase_stamps.txt).  Here the workgroup owns
// `units_per_wg` units of 64 consecutive stripes - several per wave - and a wave that has finished one takes the next from a
// counter in LDS: the waves the scheduler favours do more units, all sixteen end within one unit of each other.  The unit loop
// is the outermost scope: nothing lives across iterations but the unit number (scalar); the result window covers the line
// range of the whole workgroup as before.
// MEASURED (round 4, profiles/r04_unit_handout_ab.txt, same process A/B): with one unit per wave it runs exactly as the kernel
// above (0.699 / 0.698 of peak on the 8 GiB headline, 0.605 / 0.596 on a{1,300}); with stripes cut small enough to hand out
// several units per wave it is SLOWER on every config (email 1 GiB 0.606 -> 0.57-0.58, URL 1 GiB 0.598 -> 0.56-0.58, 8 GiB
// 0.698 -> 0.66-0.69): what a one-generation launch loses at its end is not an imbalance between waves that a finer hand-out
// could level.  Kept as an option (RRX_OPT_UNITS_PER_WORKGROUP), off by default.
template <class PhaseHook = NoPhaseHook>
__device__ __forceinline__ void dfa2_units_body(const Dfa2Device &prog, const uint8_t *__restrict__ bytes, size_t nbytes, uint32_t stripe,
                                                const uint64_t *__restrict__ stripe_base, uint32_t *__restrict__ accept_bits, uint32_t units_per_wg,
                                                PhaseHook phase = PhaseHook()) {
    phase(kPhaseEntry);
    __shared__ __attribute__((aligned(16))) struct {
        uint8_t t2_and_stage[kDfa2RegionBytes];
        uint16_t p[kDfa2PBytes / 2];
    } lds;
    Dfa2 eng;
    eng.load(prog, lds.p, lds.t2_and_stage);
    const uint32_t stage_off = (uint32_t)((Dfa2::lds_bytes(prog) + 15) & ~(size_t)15);
    uint32_t *const stage = reinterpret_cast<uint32_t *>(lds.t2_and_stage + stage_off);
    const uint32_t stage_words = (kDfa2RegionBytes - stage_off) / 4 - 1;                  // the last word of the region is the unit counter
    uint32_t &next_unit = stage[stage_words];
    for (uint32_t i = threadIdx.x; i < stage_words; i += kThreads) stage[i] = 0;
    if (threadIdx.x == 0) next_unit = kThreads / 64;                 // the first unit of every wave is its own number
    __syncthreads();
    phase(kPhaseTablesLoaded);
    const size_t g_wg = (size_t)blockIdx.x * units_per_wg * 64;      // the workgroup's first stripe (exists: the grid is sized that way)
    const uint64_t window_word = line_of(stripe_base[g_wg]) >> 5;
    uint32_t unit = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    while (unit < units_per_wg) {
        // (the lane number is made anew in every turn, by an instruction the compiler will not hoist: one more register live
        // across the stripe body and the allocator puts a round's load address into the registers its last load writes - a
        // wait state inside the burst of eight loads, which then no longer merge into one request per line: -13 %)
        uint32_t lane;
        asm volatile("v_and_b32 %0, 63, %1" : "=v"(lane) : "v"(threadIdx.x));
        dfa2_stripe<false, PhaseHook, true>(eng, g_wg + (size_t)unit * 64 + lane, window_word, stage, stage_words, bytes, nbytes, stripe, stripe_base, accept_bits,
                           nullptr, nullptr, 0u, phase);
        uint32_t ticket = 0;
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) ticket = atomicAdd(&next_unit, 1u);
        unit = __builtin_amdgcn_readfirstlane(ticket);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < stage_words; i += kThreads) {
        const uint32_t v = stage[i];
        if (v) atomicOr(&accept_bits[window_word + i], v);
    }
    phase(kPhaseWindowOut);
}
// (waves_per_eu: left to itself the allocator takes 65 registers here - one workgroup per CU instead of two)
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void match_units2_kernel(
    Dfa2Device prog, const uint8_t *__restrict__ bytes, size_t nbytes, uint32_t stripe, const uint64_t *__restrict__ stripe_base,
    uint32_t *__restrict__ accept_bits, uint32_t units_per_wg) {
    dfa2_units_body(prog, bytes, nbytes, stripe, stripe_base, accept_bits, units_per_wg);
}
__global__ __launch_bounds__(kThreads) void match_stripes2_kernel(Dfa2Device prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                   uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                                   uint32_t *__restrict__ accept_bits, uint32_t flush_mask) {
    dfa2_body<false>(prog, bytes, nbytes, stripe, stripe_base, accept_bits, nullptr, nullptr, NoPhaseHook(), flush_mask);
}
// two result bits per line (accepted, escaped) into a bitmap of twice the size: the sampled-table engine's first pass
__global__ __launch_bounds__(kThreads) void match_stripes2_two_bit_kernel(Dfa2Device prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                           uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                                           uint32_t *__restrict__ wide_bits) {
    dfa2_body<false, NoPhaseHook, 2>(prog, bytes, nbytes, stripe, stripe_base, wide_bits, nullptr, nullptr);
}
__global__ __launch_bounds__(kThreads) void match_stripes2_onepass_kernel(Dfa2Device prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                           uint32_t stripe, uint32_t *__restrict__ counts,
                                                                           uint32_t *__restrict__ slabs) {
    dfa2_body<true>(prog, bytes, nbytes, stripe, nullptr, nullptr, counts, slabs);
}

// One-pass mode, last step: lane = stripe.  The stream of stripe g (counts[g] results, the first of them dropped if the
// stripe starts inside a line) goes to bits [base, base + n) of the accept bitmap, base = '\n' before the stripe.  The
// streams of a workgroup's 256 stripes cover one contiguous bit range: they are merged in an LDS window first and leave as
// whole words, consecutive lanes writing consecutive words (atomics only because the first and the last word are shared
// with the neighbouring workgroups; words beyond the window go to memory directly).
constexpr uint32_t kCompactWindowWords = 8192;
__global__ __launch_bounds__(256) void compact_streams_kernel(const uint32_t *__restrict__ counts, const uint64_t *__restrict__ stripe_base,
                                                               size_t nstripes, uint32_t stripe, const uint32_t *__restrict__ slabs,
                                                               uint32_t *__restrict__ accept_bits, size_t cap_words) {
    __shared__ uint32_t window[kCompactWindowWords];
    const size_t g0 = (size_t)blockIdx.x * 256;
    const size_t g1 = g0 + 256 < nstripes ? g0 + 256 : nstripes;
    const uint64_t window_word = line_of(stripe_base[g0]) >> 5;          // g0 < nstripes: the grid is sized that way
    // the words this workgroup's streams can reach: up to the line the next workgroup starts in, one more for the
    // extra result of the last stripe (stripe_base has nstripes + 1 entries); only those are cleared and written out
    const uint64_t span = (line_of(stripe_base[g1]) >> 5) - window_word + 2;
    const uint32_t used = span < kCompactWindowWords ? (uint32_t)span : kCompactWindowWords;
    for (uint32_t i = threadIdx.x; i < used; i += 256) window[i] = 0;
    __syncthreads();
    const size_t g = g0 + threadIdx.x;
    if (g < nstripes) {
        const uint32_t c = counts[g];
        const uint32_t n = (c & kCountMask) + ((c & kExtraResult) ? 1u : 0u);
        const uint64_t b = stripe_base[g];
        const uint64_t base = line_of(b);
        const bool fresh = (b & kFreshStripe) != 0;
        const size_t row = (nstripes + kThreads - 1) / kThreads * kThreads;     // slab[k][stripe], rows padded to whole workgroups
        const uint32_t *src = slabs + g;
        auto put = [&](uint64_t word, uint32_t v) {
            if (!v) return;
            if (word >= cap_words) return;                          // the caller's bitmap is too small: reported from the line count
            const uint64_t rel = word - window_word;
            if (rel < used) atomicOr(&window[(uint32_t)rel], v);
            else atomicOr(&accept_bits[word], v);
        };
        for (uint32_t k = 0; k * 32 < n; k++) {
            uint32_t v = src[(size_t)k * row];
            if (n - k * 32 < 32) v &= (1u << (n - k * 32)) - 1u;
            if (k == 0 && !fresh) v &= ~1u;                          // that line belongs to the lane before me
            const uint64_t bit = base + (uint64_t)k * 32;
            const uint32_t sh = (uint32_t)bit & 31u;
            put(bit >> 5, v << sh);
            if (sh) put((bit >> 5) + 1, v >> (32u - sh));
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < used; i += 256) {
        const uint32_t v = window[i];
        if (v && window_word + i < cap_words) atomicOr(&accept_bits[window_word + i], v);
    }
}
__global__ void mail_results_kernel(const uint64_t *__restrict__ total, const uint32_t *__restrict__ flags, const uint8_t *__restrict__ last_byte,
                                    uint64_t *__restrict__ mail) {
    if (threadIdx.x || blockIdx.x) return;
    mail[0] = line_of(*total);
    mail[1] = flags ? *flags : 0u;
    mail[2] = last_byte ? *last_byte : (uint64_t)'\n';
    __threadfence_system();
}

// ============================================================================================ line index
// counts[g] = number of '\n' in stripe g, streamed exactly like the match kernel streams it.  Also raises
// *flags bit 0 if any byte >= 0x80 occurs (the match kernel then clamps such bytes to the dead column).
__global__ __launch_bounds__(256) void count_newlines_kernel(const uint8_t *__restrict__ bytes, size_t nbytes, uint32_t stripe,
                                                              uint32_t *__restrict__ counts, size_t nstripes,
                                                              uint32_t *__restrict__ flags) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= nstripes) return;
    const size_t start = g * (size_t)stripe;
    const size_t end = start + stripe < nbytes ? start + stripe : nbytes;
    const uint4 *src = reinterpret_cast<const uint4 *>(bytes + start);
    const int units = (int)((end - start) / 16);
    uint32_t cnt = 0, high = 0;
    int u = 0;
    for (; u + 4 <= units; u += 4) {       // 64-byte bursts: with next to no work per byte this is the fastest feed
        uint4 v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = src[u + i];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t x = w[j] ^ 0x0a0a0a0au;                                        // zero byte <=> '\n'
                uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);    // exact zero-byte test
                cnt += __popc(z);
                high |= w[j];
            }
        }
    }
    for (size_t p = start + (size_t)u * 16; p < end; p++) { cnt += bytes[p] == '\n'; high |= bytes[p]; }
    // bit 31: the stripe ends on a '\n', i.e. the next stripe starts a fresh line (the scan moves it to bit 63 of
    // that stripe's base, so the match kernels need not probe the byte before their stripe)
    counts[g] = cnt | (bytes[end - 1] == '\n' ? kEndsOnNewline : 0u);
    if (high & 0x80808080u) atomicOr(flags, 1u);
}

// The same counts, a WAVE per stripe: lane l reads 16 bytes at l*16 of every KiB of the stripe, so a wave instruction
// reads one contiguous KiB (the lane-per-stripe kernel above reads like the match kernel does, 64 lines 64 stripes apart
// per instruction, and reaches 4.8 TB/s; nothing here has to agree with the match kernel's geometry but the counts).
// Stripes are multiples of 1 KiB; the corpus' last, partial stripe is counted byte by byte.
__global__ __launch_bounds__(256) void count_newlines_wave_kernel(const uint8_t *__restrict__ bytes, size_t nbytes, uint32_t stripe,
                                                                   uint32_t *__restrict__ counts, size_t nstripes,
                                                                   uint32_t *__restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const size_t nwaves = (size_t)gridDim.x * 4;
    uint32_t high = 0;
    for (size_t g = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6; g < nstripes; g += nwaves) {
        const size_t start = g * (size_t)stripe;
        const size_t end = start + stripe < nbytes ? start + stripe : nbytes;
        uint32_t cnt = 0, last = 0;
        if (end - start == stripe) {
            const uint4 *src = reinterpret_cast<const uint4 *>(bytes + start) + lane;
            const int n = (int)(stripe >> 10);                       // KiB per stripe: 1, 2, 4, 8, 16
            for (int i = 0; i < n; i += 4) {
                uint4 v[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (i + k < n) v[k] = src[(size_t)(i + k) * 64];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (i + k < n) {
                        const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t x = w[j] ^ 0x0a0a0a0au;                                        // zero byte <=> '\n'
                            const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);    // exact zero-byte test
                            cnt += __popc(z);
                            high |= w[j];
                        }
                        last = v[k].w >> 24;                         // (lane 63 of the last KiB: the stripe's last byte)
                    }
                }
            }
            last = __shfl(last, 63, 64);
        } else {
            for (size_t p = start + lane; p < end; p += 64) { const uint32_t b = bytes[p]; cnt += b == '\n'; high |= b; }
            last = bytes[end - 1];
        }
#pragma unroll
        for (int d = 32; d; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
        if (lane == 0) counts[g] = cnt | (last == '\n' ? kEndsOnNewline : 0u);
    }
    if (__ballot((high & 0x80808080u) != 0) && lane == 0) atomicOr(flags, 1u);
}

// bytes[i] = bit i of the accept bitmap (the byte-per-line form of the result)
__global__ __launch_bounds__(256) void expand_bits_kernel(const uint32_t *__restrict__ bits, size_t nlines, uint8_t *__restrict__ out) {
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;          // one 32-line word -> 32 bytes
    if (w * 32 >= nlines) return;
    const uint32_t v = bits[w];
    if (w * 32 + 32 <= nlines) {
        uint4 o[2];
        uint32_t *p = reinterpret_cast<uint32_t *>(o);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t n = (v >> (4 * j)) & 0xfu;
            p[j] = (n & 1u) | ((n & 2u) << 7) | ((n & 4u) << 14) | ((n & 8u) << 21);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(out + w * 32);
        dst[0] = o[0]; dst[1] = o[1];
    } else {
        for (size_t i = w * 32; i < nlines; i++) out[i] = (uint8_t)((v >> (i & 31)) & 1u);
    }
}

// exclusive scan of n counts into n+1 bases, two levels: (1) every workgroup sums its chunk of kScanChunk
// counts; (2) one workgroup scans the chunk sums; (3) every workgroup scans its chunk from its chunk base.
constexpr int kScanChunk = 4096;
__global__ __launch_bounds__(256) void scan_chunk_sums_kernel(const uint32_t *__restrict__ counts, size_t n, uint64_t *__restrict__ sums) {
    __shared__ uint64_t part[4];
    const size_t lo = (size_t)blockIdx.x * kScanChunk;
    uint64_t s = 0;
    for (size_t i = lo + threadIdx.x; i < lo + kScanChunk && i < n; i += 256) s += counts[i] & kCountMask;
#pragma unroll
    for (int d = 32; d; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// exclusive prefix over the workgroup of one value per thread (NW waves): shuffles inside a wave, the NW wave totals
// through LDS.  (The first version let thread 0 walk the partial sums one by one: 10-50 us per scan kernel, as much as
// the one-shot entry's compaction.)
template <int NW>
__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t *wave_tot, uint64_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint64_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const uint64_t t = wave_tot[w];
        if (w < wave) off += t;
        tot += t;
    }
    total = tot;
    return off + inc - v;
}
__global__ __launch_bounds__(1024) void scan_sums_kernel(uint64_t *__restrict__ sums, size_t nchunks, uint64_t *__restrict__ total) {
    __shared__ uint64_t wave_tot[16];
    const size_t per = (nchunks + 1023) / 1024;
    const size_t lo = threadIdx.x * per < nchunks ? threadIdx.x * per : nchunks, hi = lo + per < nchunks ? lo + per : nchunks;
    uint64_t s = 0;
    for (size_t i = lo; i < hi; i++) s += sums[i];
    uint64_t all;
    uint64_t run = block_exclusive_scan<16>(s, wave_tot, all);
    if (threadIdx.x == 0) *total = all;
    for (size_t i = lo; i < hi; i++) { uint64_t v = sums[i]; sums[i] = run; run += v; }
}
__global__ __launch_bounds__(256) void scan_chunks_kernel(const uint32_t *__restrict__ counts, size_t n, const uint64_t *__restrict__ sums,
                                                           uint64_t *__restrict__ base) {
    __shared__ uint64_t wave_tot[4];
    constexpr int kPer = kScanChunk / 256;
    const size_t lo = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kPer;
    uint32_t c[kPer];
    uint32_t before = 0;                                   // the count word in front of mine (its kEndsOnNewline flag)
    if (lo + kPer <= n) {
        const uint4 *src = reinterpret_cast<const uint4 *>(counts + lo);      // lo is a multiple of kPer = 16 words
#pragma unroll
        for (int k = 0; k < kPer / 4; k++) { const uint4 v = src[k]; c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w; }
    } else {
#pragma unroll
        for (int k = 0; k < kPer; k++) c[k] = lo + k < n ? counts[lo + k] : 0u;
    }
    if (lo && lo < n) before = counts[lo - 1];
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) s += c[k] & kCountMask;
    uint64_t all;
    uint64_t run = sums[blockIdx.x] + block_exclusive_scan<4>(s, wave_tot, all);
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        if (lo + k < n) {
            const bool fresh = lo + k == 0 || ((k ? c[k - 1] : before) & kEndsOnNewline);
            base[lo + k] = run | (fresh ? kFreshStripe : 0);
        }
        run += c[k] & kCountMask;
    }
}

// ============================================================================================ search: patterns that accept ""
// The stripe-wise kernels (kernels_search.hip) serve every pattern that does not accept the empty string.  One that does has
// a match [k, k) at EVERY offset k = 0 .. length of its line (the search moves on by one byte after an empty match), whatever
// the text: no table, only the line lengths.  line_offsets_kernel (once per corpus): lane = stripe, every '\n' at p inside the
// stripe starts the next line at p + 1 (line numbers from the stripe index).  empty_matches_kernel: lane = line.
__global__ __launch_bounds__(256) void line_offsets_kernel(const uint8_t *__restrict__ bytes, size_t nbytes, uint32_t stripe,
                                                           const uint64_t *__restrict__ stripe_base, size_t nstripes,
                                                           uint64_t *__restrict__ line_off) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nstripes) return;
    const size_t start = g * (size_t)stripe, end = start + stripe < nbytes ? start + stripe : nbytes;
    uint64_t line = line_of(stripe_base[g]);             // index of the line that contains my first byte
    if (g == 0) line_off[0] = 0;
    size_t pos = start;
    for (; pos + 16 <= end; pos += 16) {                 // stripes start 16-byte aligned
        const uint4 v = *reinterpret_cast<const uint4 *>(bytes + pos);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t x = w[j] ^ 0x0a0a0a0au;
            uint32_t hit = (x - 0x01010101u) & ~x & 0x80808080u;        // exact for the lowest flagged byte; refined below
            while (hit) {
                const int k = (__ffs((int)hit) - 1) >> 3;
                if (((w[j] >> (8 * k)) & 0xffu) == '\n') line_off[++line] = pos + 4 * j + k + 1;
                hit &= hit - 1;
            }
        }
    }
    for (; pos < end; pos++)
        if (bytes[pos] == '\n') line_off[++line] = pos + 1;
}

// FILL = false: count[i] = length of line i + 1.  FILL = true: the matches of line i go to the slots first[i], first[i] + 1, ...
template <bool FILL>
__global__ __launch_bounds__(256) void empty_matches_kernel(const uint64_t *__restrict__ line_off, size_t nlines, uint32_t *__restrict__ count,
                                                            const uint64_t *__restrict__ first, uint32_t *__restrict__ match_start,
                                                            uint32_t *__restrict__ match_end, uint64_t cap) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlines) return;
    const uint64_t len = line_off[i + 1] - 1 - line_off[i];      // the line without its '\n'
    if (!FILL) { count[i] = (uint32_t)(len + 1); return; }
    const uint64_t slot = first[i];
    for (uint64_t k = 0; k <= len && slot + k < cap; k++) { match_start[slot + k] = (uint32_t)k; match_end[slot + k] = (uint32_t)k; }
}

// ============================================================================================ explicit items, stripe-wise
// (r4) The item-end bitmap is stored TRANSPOSED inside groups of 64 stripes: the 16 bytes (128 marks) of stripe g's round r sit at
// ((g / 64) * rounds + r) * 1 KiB + (g % 64) * 16 - what the 64 lanes of a wave ask for in one round is one contiguous KiB.  In the
// plain order a lane's 16 bytes lay stripe / 8 bytes from its neighbour's, a cache line each, and by the lane's next round the
// line was gone again: the items kernels fetched 2.2 x the bytes of the text (FETCH_SIZE; 0.36 ms per GiB whatever the table).
// A permutation of 16-byte pieces inside a group's part of the bitmap: the index kernel writes every word once as before.
__device__ __forceinline__ size_t ends_slot(size_t word, uint32_t sw_log2) {           // sw_log2 = log2(stripe / 32): words per stripe
    const size_t g = word >> sw_log2;
    const uint32_t j = (uint32_t)word & ((1u << sw_log2) - 1u);
    return ((((g >> 6) << (sw_log2 - 2)) + (j >> 2)) << 8) + ((g & 63) << 2) + (j & 3);
}
// rrx_match_extents on a large batch (an offsets array over one byte buffer: an Arrow-style string column): the items are
// lines without a delimiter.  match_extents_kernel gives every lane an item (0.9-1.0 TB/s: consecutive lanes read text an
// item apart).  Here the buffer is cut into stripes exactly like a corpus, and the item ends come from a bitmap built from
// the offsets (1 bit per byte) instead of a byte value.  The table is the plain table in the wide line-table format with
// one more column (abi.cpp: items_table): byte values 0..127 - '\n' an ordinary byte -, 128 = any byte >= 0x80, 129 = END OF
// ITEM (the verdict of the row, back to the start row):
//   ENDS = 1 (trim 1: every item is followed by one separator byte): the marked byte is the separator, stepped as byte 129;
//   ENDS = 2 (trim 0): the marked byte is the item's last byte, a byte 129 is stepped after it.
// Bytes >= 0x80 of the text are stepped as 0x80.
template <int ENDS>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void match_items_stripes_kernel(LineDfaDevice prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                        uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                                        const uint32_t *__restrict__ ends, uint32_t *__restrict__ accept_bits,
                                                                        uint32_t stage_off, uint32_t stage_words,
                                                                        const uint64_t *__restrict__ off, size_t nitems,
                                                                        const uint32_t *__restrict__ skip_if) {
    typedef LineDfaEngine<true, false> Engine;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // One-call form (rrx_match_extents, asynchronous): the host knows neither where the batch starts nor how long it is -
    // both come from the offsets here - and the index pass may have found the batch unfit (*skip_if != 0: an item without
    // a byte for its mark, a misaligned start, too small): then the lane-per-item kernel queued behind this one runs instead.
    if (skip_if && *skip_if) return;
    if (off) { const uint64_t first = off[0]; bytes += first; nbytes = (size_t)(off[nitems] - first); }
    if ((size_t)blockIdx.x * kThreads * stripe >= nbytes) return;    // (the grid was sized from an upper bound)
    uint32_t *stage = reinterpret_cast<uint32_t *>(smem + stage_off);
    Engine eng;
    eng.load(prog, smem);
    for (uint32_t i = threadIdx.x; i < stage_words; i += kThreads) stage[i] = 0;
    __syncthreads();
    const size_t g0 = (size_t)blockIdx.x * kThreads;
    const uint64_t window_word = line_of(stripe_base[g0]) >> 5;
    const size_t g = g0 + threadIdx.x;
    const size_t start = g * (size_t)stripe;
    if (start < nbytes) {
        const size_t stripe_end = start + stripe;
        const size_t my_end = stripe_end < nbytes ? stripe_end : nbytes;
        const uint64_t my_base = stripe_base[g];
        const bool fresh = (my_base & kFreshStripe) != 0;
        ResultsT<true> res;
        res.begin_staged(line_of(my_base), window_word, !fresh, accept_bits, stage);
        res.stage_words = stage_words;
        typename Engine::State st = fresh ? eng.fresh() : eng.skipping();
        const uint32_t swl = (uint32_t)__builtin_ctz(stripe) - 5u;
        auto end_bit = [&](size_t pos) -> bool { return (ends[ends_slot(pos >> 5, swl)] >> (pos & 31)) & 1u; };
        // one byte with its end bit: -> (nl, acc) of the step that matters
        // (plain entries carry "the next row is accepting" in bit 7 of the byte that is the line count elsewhere: see word())
        auto step1 = [&](uint32_t c, uint32_t &nl, uint32_t &acc) { eng.step(st, c, nl, acc); nl &= 1u; };
        auto step_byte = [&](size_t pos, uint32_t &nl, uint32_t &acc) {
            uint32_t c = bytes[pos];
            const bool m = end_bit(pos);
            if (c >= 0x80u) c = 0x80u;
            if (ENDS == 1 && m) c = kItemEndColumn;
            step1(c, nl, acc);
            if (ENDS == 2 && m) step1(kItemEndColumn, nl, acc);
        };
        // a text word (no byte >= 0x81 in it) with the end bits m4 of its four bytes
        auto word = [&](uint32_t w, uint32_t m4) {
            if constexpr (ENDS == 1) {
                {   // (no test for "some lane has a separator in this word": with 64 lanes it is nearly always so)
                    // bit k of m4 -> byte k (24-bit multiply: v_mul_lo_u32 runs at a quarter of the rate)
                    const uint32_t t = __umul24(m4, 0x00204081u) & 0x01010101u;
                    const uint32_t bm = (t << 8) - t;
                    w = (w & ~bm) | (0x81818181u & bm);
                }
                eng.consume_word(st, w, res.bits);
            } else {
                // trim 0: an item that ends ON this byte reports the verdict of the row the byte leads to and goes back to the start row
                // - the END column's entry, whose verdict the plain entry carries in bit 23.  So the marked lanes take (start row | one
                // line | that verdict) in place of what they read: three VALU more per byte, no second lookup, no branch.  (Round 2 and
                // the first half of round 3 stepped the END column under a wave-wide test per byte: with 64 lanes some lane nearly always
                // has a mark, so nearly every byte paid two dependent lookups - 12.7 VALU, 5.4 SALU and 1.8 LDS reads per byte.)
                const uint32_t end_entry = eng.start_off | 1u << 16;
                uint32_t mk, x, t0, t1;
#define RRX_ITEM_BYTE(SEL, KBIT)                                                                                                          \
                asm volatile("v_lshlrev_b32_sdwa %[c4], %[two], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" SEL "\n\t" \
                             "v_add_u32_sdwa %[ad], %[e], %[c4] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"       \
                             "ds_read_b32 %[e], %[ad]\n\t"                                                                                  \
                             "v_bfe_i32 %[mk], %[m4], " KBIT ", 1\n\t"                                                                       \
                             "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
                             "v_and_b32 %[x], 0x800000, %[e]\n\t"                                                                           \
                             "v_lshl_or_b32 %[x], %[x], 1, %[ee]\n\t"                                                                       \
                             "v_bfi_b32 %[e], %[mk], %[x], %[e]\n\t"                                                                        \
                             "v_lshlrev_b32_sdwa %[b], %[e], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n\t"     \
                             "v_or_b32_sdwa %[b], %[e], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD"              \
                             : [e] "+v"(st.e), [b] "+v"(res.bits), [c4] "=&v"(t0), [ad] "=&v"(t1), [mk] "=&v"(mk), [x] "=&v"(x)               \
                             : [w] "v"(w), [two] "v"(eng.col_shift), [m4] "v"(m4), [ee] "v"(end_entry)                                       \
                             : "memory")
                RRX_ITEM_BYTE("BYTE_0", "0"); RRX_ITEM_BYTE("BYTE_1", "1"); RRX_ITEM_BYTE("BYTE_2", "2"); RRX_ITEM_BYTE("BYTE_3", "3");
#undef RRX_ITEM_BYTE
            }
        };
        auto clamp = [](uint32_t w) -> uint32_t { const uint32_t hi = w & 0x80808080u; return w & ~(hi - (hi >> 7)); };      // >= 0x80 -> 0x80
        size_t pos = start;
        const uint4 *src = reinterpret_cast<const uint4 *>(bytes + start);
        const uint4 *esrc = reinterpret_cast<const uint4 *>(ends + ends_slot(start >> 5, swl));      // 128 bits per 128-byte round, rounds 1 KiB apart
        constexpr int kSlots = kRound / 16;
        const int rounds = (int)((my_end - start) / kRound);
        TextRound<kSlots> buf;
        uint4 eb = make_uint4(0, 0, 0, 0);
        if (rounds > 0) { buf.load(src); eb = esrc[0]; }
        for (int r = 0; r < rounds; r++) {
            const uint32_t ew[4] = {eb.x, eb.y, eb.z, eb.w};
            int slot = 0;
            buf.for_each_slot([&](const uint4 &v) {
                const uint32_t sb = (ew[slot >> 1] >> (16 * (slot & 1))) & 0xffffu;           // (slot: a constant after inlining)
                if (__builtin_amdgcn_ballot_w64(((v.x | v.y | v.z | v.w) & 0x80808080u) != 0)) {     // rare on text: one test per 16 bytes
                    word(clamp(v.x), sb & 15u); word(clamp(v.y), (sb >> 4) & 15u); word(clamp(v.z), (sb >> 8) & 15u); word(clamp(v.w), sb >> 12);
                } else {
                    word(v.x, sb & 15u); word(v.y, (sb >> 4) & 15u); word(v.z, (sb >> 8) & 15u); word(v.w, sb >> 12);
                }
                if (res.bits >> 15) res.flush();
                slot++;
            });
            if ((r & 3) == 3) res.flush();
            if (r + 1 < rounds) { buf.load(src + (size_t)(r + 1) * kSlots); eb = esrc[(size_t)(r + 1) * 64]; }
        }
        pos += (size_t)rounds * kRound;
        for (; pos < my_end; pos++) {                                 // tail of the buffer inside my stripe
            uint32_t nl, acc;
            step_byte(pos, nl, acc);
            res.push(nl, acc);
            if (res.bits >> 30) res.flush();
        }
        res.flush();
        // the item that straddles my stripe end is mine if it started here: follow it to its end
        const bool started = fresh || res.seen > 0;
        if (started && !end_bit(my_end - 1)) {
            uint32_t nl = 0, acc = 0;
            // 16 bytes and their 16 end bits per turn (pos is 16-byte aligned: stripes are multiples of 128; one byte and one
            // bitmap word per turn was a chain of 150 memory round trips for the slowest lane of a wave on 95-byte items)
            while (pos + 16 <= nbytes && !nl) {
                const uint4 v = *reinterpret_cast<const uint4 *>(bytes + pos);
                const uint32_t e16 = (ends[ends_slot(pos >> 5, swl)] >> (pos & 31)) & 0xffffu;
                const uint32_t w[4] = {clamp(v.x), clamp(v.y), clamp(v.z), clamp(v.w)};
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (!nl) {
                        uint32_t c = (w[k >> 2] >> (8 * (k & 3))) & 0xffu;
                        const bool m = (e16 >> k) & 1u;
                        if (ENDS == 1 && m) c = kItemEndColumn;
                        step1(c, nl, acc);
                        if (ENDS == 2 && m) step1(kItemEndColumn, nl, acc);
                    }
                }
                pos += 16;
            }
            for (; pos < nbytes && !nl; pos++) step_byte(pos, nl, acc);
            if (!nl) step1(kItemEndColumn, nl, acc);                   // (cannot happen: the last item ends where the buffer ends)
            res.push(nl, acc);
        }
        res.finish();
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < stage_words; i += kThreads) {
        const uint32_t v = stage[i];
        if (v) atomicOr(&accept_bits[window_word + i], v);
    }
}
// (r4) The same batch with a separator byte behind every item (trim 1) on the STRIDE-2 table of its own (lower_dfa2's items form):
// codes 0 ... 127 are the byte values - '\n' an ordinary byte -, code 128 is END OF ITEM, and the kernel puts it in the place of
// every marked byte (one v_perm_b32 per text word, its selector made from the word's four mark bits); bytes >= 0x80 are stepped
// as 0x00, which no pattern takes either.  From there on it is the batch kernel's step - two bytes per dependent lookup - with the
// items kernel's stripes, marks and result window.  (trim 0 stays on the byte-stride kernel above: an item that ends ON a byte
// needs that byte and the end in one symbol, and a pair with a mark on its first byte a second dependent lookup.)
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void match_items_stripes2_kernel(Dfa2Device prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                         uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                                         const uint32_t *__restrict__ ends, uint32_t *__restrict__ accept_bits,
                                                                         const uint64_t *__restrict__ off, size_t nitems,
                                                                         const uint32_t *__restrict__ skip_if) {
    __shared__ __attribute__((aligned(16))) struct {
        uint8_t t2_and_stage[kDfa2RegionBytes];
        uint16_t p[kDfa2PItemsBytes / 2];
    } lds;
    if (skip_if && *skip_if) return;                                 // (see match_items_stripes_kernel)
    if (off) { const uint64_t first = off[0]; bytes += first; nbytes = (size_t)(off[nitems] - first); }
    if ((size_t)blockIdx.x * kThreads * stripe >= nbytes) return;
    Dfa2 eng;
    eng.load(prog, lds.p, lds.t2_and_stage, kDfa2PItemsBytes);
    const uint32_t stage_off = (uint32_t)((Dfa2::lds_bytes(prog) + 15) & ~(size_t)15);
    uint32_t *const stage = reinterpret_cast<uint32_t *>(lds.t2_and_stage + stage_off);
    const uint32_t stage_words = (kDfa2RegionBytes - stage_off) / 4;
    for (uint32_t i = threadIdx.x; i < stage_words; i += kThreads) stage[i] = 0;
    __syncthreads();
    const size_t g0 = (size_t)blockIdx.x * kThreads;
    const uint64_t window_word = line_of(stripe_base[g0]) >> 5;
    const size_t g = g0 + threadIdx.x;
    const size_t start = g * (size_t)stripe;
    if (start < nbytes) {
        const size_t stripe_end = start + stripe;
        const size_t my_end = stripe_end < nbytes ? stripe_end : nbytes;
        const uint64_t my_base = stripe_base[g];
        const bool fresh = (my_base & kFreshStripe) != 0;
        ResultsT<true> res;
        res.begin_staged(line_of(my_base), window_word, !fresh, accept_bits, stage);
        res.stage_words = stage_words;
        Dfa2::State st = fresh ? eng.fresh() : eng.skipping();
        const uint32_t swl = (uint32_t)__builtin_ctz(stripe) - 5u;
        auto end_bit = [&](size_t pos) -> bool { return (ends[ends_slot(pos >> 5, swl)] >> (pos & 31)) & 1u; };
        auto code_at = [&](size_t pos) -> uint32_t { const uint32_t c = bytes[pos]; return end_bit(pos) ? 128u : c >= 0x80u ? 0u : c; };
        auto clean = [](uint32_t w) -> uint32_t { const uint32_t hi = (w & 0x80808080u) >> 7; return w & ~(hi * 0xffu); };       // >= 0x80 -> 0x00
        // the four bytes of a text word (none >= 0x80) with their mark bits m4: marked bytes become code 128
        auto word = [&](uint32_t w, uint32_t m4) {
            // bit k of m4 -> bit 2 of byte k: selector k + 4 (a byte of the constant) where marked, k (the text byte) elsewhere
            const uint32_t sel = (__umul24(m4, 0x00810204u) & 0x04040404u) | 0x03020100u;
            eng.consume_dword_items(st, __builtin_amdgcn_perm(0x80808080u, w, sel), res.bits);
        };
        size_t pos = start;
        const uint4 *src = reinterpret_cast<const uint4 *>(bytes + start);
        const uint4 *esrc = reinterpret_cast<const uint4 *>(ends + ends_slot(start >> 5, swl));      // 128 bits per 128-byte round, rounds 1 KiB apart
        constexpr int kSlots = kRound / 16;
        const int rounds = (int)((my_end - start) / kRound);
        TextRound<kSlots> buf;
        uint4 eb = make_uint4(0, 0, 0, 0);
        if (rounds > 0) { buf.load(src); eb = esrc[0]; }
        for (int r = 0; r < rounds; r++) {
            const uint32_t ew[4] = {eb.x, eb.y, eb.z, eb.w};
            int slot = 0;
            buf.for_each_slot([&](const uint4 &v) {
                const uint32_t sb = (ew[slot >> 1] >> (16 * (slot & 1))) & 0xffffu;           // (slot: a constant after inlining)
                if (__builtin_amdgcn_ballot_w64(((v.x | v.y | v.z | v.w) & 0x80808080u) != 0)) {     // rare on text: one test per 16 bytes
                    word(clean(v.x), sb & 15u); word(clean(v.y), (sb >> 4) & 15u); word(clean(v.z), (sb >> 8) & 15u); word(clean(v.w), sb >> 12);
                } else {
                    word(v.x, sb & 15u); word(v.y, (sb >> 4) & 15u); word(v.z, (sb >> 8) & 15u); word(v.w, sb >> 12);
                }
                if (res.bits >> 15) res.flush();
                slot++;
            });
            if ((r & 3) == 3) res.flush();
            if (r + 1 < rounds) { buf.load(src + (size_t)(r + 1) * kSlots); eb = esrc[(size_t)(r + 1) * 64]; }
        }
        pos += (size_t)rounds * kRound;
        // tail of the buffer inside my stripe (only the last stripe has one): whole pairs, then an odd last byte paired with a
        // virtual END.  The batch's last byte is its last item's separator: marked, so the odd byte reports two ends of which only
        // the first exists.
        for (; pos + 2 <= my_end; pos += 2) {
            uint32_t lines, verdicts;
            eng.step2(st, code_at(pos), code_at(pos + 1), lines, verdicts);
            res.bits = (res.bits << lines) | verdicts;
            if (res.bits >> 29) res.flush();
        }
        bool closed_by_end_of_data = false;
        if (pos < my_end) {
            const uint32_t c = code_at(pos);
            uint32_t lines, verdicts;
            eng.step2(st, c, 128u, lines, verdicts);
            if (c == 128u) res.push(1, verdicts >> 1);
            else { res.push(1, verdicts); closed_by_end_of_data = true; }       // (cannot happen: see above)
            pos++;
        }
        res.flush();
        // the item that straddles my stripe end is mine if it started here: follow it to its end, pair by pair (stripes are even-sized)
        const bool started = fresh || res.seen > 0;
        if (!closed_by_end_of_data && started && !end_bit(my_end - 1)) {
            uint32_t lines = 0, verdicts = 0;
            while (pos + 16 <= nbytes && !lines) {
                const uint4 v = *reinterpret_cast<const uint4 *>(bytes + pos);
                const uint32_t e16 = (ends[ends_slot(pos >> 5, swl)] >> (pos & 31)) & 0xffffu;
                const uint32_t w[4] = {clean(v.x), clean(v.y), clean(v.z), clean(v.w)};
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (!lines) {
                        const uint32_t c1 = (e16 >> (2 * k)) & 1u ? 128u : (w[k >> 1] >> (16 * (k & 1))) & 0xffu;
                        const uint32_t c2 = (e16 >> (2 * k + 1)) & 1u ? 128u : (w[k >> 1] >> (16 * (k & 1) + 8)) & 0xffu;
                        eng.step2(st, c1, c2, lines, verdicts);
                    }
                }
                pos += 16;
            }
            for (; pos + 2 <= nbytes && !lines; pos += 2) eng.step2(st, code_at(pos), code_at(pos + 1), lines, verdicts);
            if (!lines) eng.step2(st, pos < nbytes ? code_at(pos) : 128u, 128u, lines, verdicts);
            res.push(1, lines == 2 ? verdicts >> 1 : verdicts);               // only the first end of the pair is mine
        }
        res.finish();
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < stage_words; i += kThreads) {
        const uint32_t v = stage[i];
        if (v) atomicOr(&accept_bits[window_word + i], v);
    }
}
// The index of a batch of items from its offsets, in ONE pass over them (round 3; round 2: a memset of the bitmap, a kernel for
// the item ends with atomics on the words two workgroups share, and a binary search per stripe):
//  * ends: bitmap of the item ends (positions relative to off[0]); *flag |= 1 if an item has no byte to carry its mark.  The
//    ends are sorted: the 1024 items of a workgroup mark a contiguous range of bitmap words, and the workgroup OWNS the words
//    [word of its first mark, word of the next workgroup's first mark) - it leaves its own last marks that fall into the next
//    owner's first word to that owner, and picks up the marks the items in front of its own left in its first word (at most 32:
//    every item has a byte of its own).  The range is assembled in LDS, tile by tile, and written out whole with plain stores,
//    zero words included: every word of the bitmap is written exactly once, nothing is cleared beforehand, nothing is atomic.
//    (One atomic per item: 0.12 ms for 22 M items of 95 bytes, 1.0 ms for 55 M of 19; one 4-byte store per marked word:
//    0.34 / 0.83 ms - scattered partial writes into 128 MB.)
//  * stripe_base[g] = items that end before stripe g | kFreshStripe if the byte in front of it is marked; entry nstripes = all.
//    The item that holds a stripe's first byte knows its own number: it writes the entry (no search, no scan); the stripes behind
//    the batch's last byte (the index is laid out for an upper bound of its extent) are filled in by everybody.
constexpr uint32_t kEndsTile = 4096, kEndsPerLane = 4, kEndsItems = 256 * kEndsPerLane;      // items per workgroup
__global__ __launch_bounds__(256) void item_index_kernel(const uint64_t *__restrict__ off, size_t nitems, uint32_t trim, uint32_t *__restrict__ ends,
                                                         uint32_t *__restrict__ flag, uint64_t limit_words, const uint8_t *__restrict__ bytes_base,
                                                         uint64_t min_bytes, uint32_t stripe_log2, size_t nstripes, uint64_t *__restrict__ stripe_base) {
    __shared__ uint32_t tile[kEndsTile];
    const uint64_t base = off[0], extent = off[nitems] > base ? off[nitems] - base : 0;
    // one-call form: the batch's extent is only known here.  Unfit (flag bit 1) if it is shorter than the stripe-wise path
    // pays for, longer than the bitmap was sized for, or does not start on a 16-byte boundary.
    if (bytes_base && blockIdx.x == 0 && threadIdx.x == 0) {
        if (!extent || extent < min_bytes || ((extent + 31) >> 5) > limit_words || (reinterpret_cast<uintptr_t>(bytes_base + base) & 15)) atomicOr(flag, 2u);
    }
    const size_t i0 = (size_t)blockIdx.x * kEndsItems;
    const size_t i1 = i0 + kEndsItems < nitems ? i0 + kEndsItems : nitems;     // first item of the next workgroup (nitems: none)
    auto mark_of = [&](size_t k) -> uint64_t {                      // position of item k's mark (a degenerate item: of its start)
        const uint64_t e = off[k + 1];
        return (e > base ? e - 1 : base) - base;
    };
    uint64_t word[kEndsPerLane];
    uint32_t mask[kEndsPerLane];
    uint64_t ob[kEndsPerLane], oe[kEndsPerLane];
#pragma unroll
    for (uint32_t k = 0; k < kEndsPerLane; k++) {                   // (all loads first: four round trips in flight)
        const size_t i = i0 + (size_t)k * 256 + threadIdx.x;
        ob[k] = i < nitems ? off[i] : 0;
        oe[k] = i < nitems ? off[i + 1] : 0;
    }
    bool degenerate = false;
    const uint64_t stripe_mask = ((uint64_t)1 << stripe_log2) - 1;
#pragma unroll
    for (uint32_t k = 0; k < kEndsPerLane; k++) {
        const size_t i = i0 + (size_t)k * 256 + threadIdx.x;
        word[k] = ~0ull; mask[k] = 0;
        if (i < nitems) {
            if (oe[k] <= ob[k] || oe[k] - ob[k] < trim) degenerate = true;     // trim 1: at least the separator; trim 0: at least one byte
            else { const uint64_t pos = oe[k] - 1 - base; word[k] = pos >> 5; mask[k] = 1u << (pos & 31); }
            if (oe[k] > ob[k] && ob[k] >= base) {                   // the stripes whose first byte is one of mine
                const uint64_t s0 = ob[k] - base, e0 = oe[k] - base;
                uint64_t g = (s0 + stripe_mask) >> stripe_log2;
                const uint64_t g1 = (e0 + stripe_mask) >> stripe_log2;
                for (; g < g1 && g < nstripes; g++)
                    stripe_base[g] = (uint64_t)i | ((g == 0 || (g << stripe_log2) == s0) ? kFreshStripe : 0);
            }
        }
    }
    if (degenerate) atomicOr(flag, 1u);
    {   // stripes that begin at or behind the batch's last byte, and the closing entry
        const uint64_t gend = (extent + stripe_mask) >> stripe_log2;
        for (uint64_t g = gend + (uint64_t)blockIdx.x * 256 + threadIdx.x; g <= nstripes; g += (uint64_t)gridDim.x * 256)
            stripe_base[g] = (uint64_t)nitems | ((g < nstripes && (g == 0 || (g << stripe_log2) == extent)) ? kFreshStripe : 0);
    }
    const uint64_t F = blockIdx.x == 0 ? 0 : mark_of(i0) >> 5;      // my words: [F, X)
    uint64_t X = i1 < nitems ? mark_of(i1) >> 5 : ((extent + 31) >> 5) + 4;
    if (X > limit_words) X = limit_words;
    for (uint64_t T = F; T < X; T += kEndsTile) {
        const uint64_t n = X - T < kEndsTile ? X - T : kEndsTile;   // words of this tile
        for (uint32_t j = threadIdx.x; j < n; j += 256) tile[j] = 0;
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < kEndsPerLane; k++)
            if (word[k] >= T && word[k] < T + n) atomicOr(&tile[(uint32_t)(word[k] - T)], mask[k]);
        if (T == F && threadIdx.x < 32 && i0 >= 1 + (size_t)threadIdx.x) {      // what the items in front of mine left in my first word
            const size_t ip = i0 - 1 - threadIdx.x;
            const uint64_t b = off[ip], e = off[ip + 1];
            if (e > b && e - b >= trim && e > base && ((e - 1 - base) >> 5) == F) atomicOr(&tile[0], 1u << ((e - 1 - base) & 31));
        }
        __syncthreads();
        {   // write-out in the order of the TRANSPOSED layout (ends_slot): the words of the tile's stripes round by round, so that
            // consecutive lanes write consecutive 16-byte pieces (in the tile's own order every piece lands a KiB from the last:
            // 111 us instead of 71 for the index of a GiB)
            const uint32_t swl = stripe_log2 - 5u, sw = 1u << swl;
            const uint64_t gA = T >> swl;
            const uint32_t nst = (uint32_t)(((T + n - 1) >> swl) - gA) + 1u;
            const uint32_t total = nst << swl;
            for (uint32_t idx = threadIdx.x; idx < total; idx += 256) {
                const uint32_t q = idx >> 2, r = q / nst, s_ = q - r * nst;
                const uint64_t src = ((gA + s_) << swl) + 4u * r + (idx & 3u);
                if (src >= T && src < T + n) ends[ends_slot(src, swl)] = tile[(uint32_t)(src - T)];
            }
            (void)sw;
        }
        __syncthreads();
    }
}

// ============================================================================================ one long string
// Chunk maps.  LDS: the plain DFA widened to one u16 entry per (state, byte value 0..127 | >= 0x80), entry = row
// offset of the next state (state * 129), so a step is one clamp, one add and one ds_read_u16.
constexpr int kLongThreads = 256;
//
// Convergence (round 2): stepping a chunk from EVERY state costs D times the text.  But a DFA forgets where it started:
// after a few dozen bytes the D runs of a chunk sit in one, two, three different states (a whole-string match against
// running text is dead almost at once).  So the chunk maps are built in four steps:
//   A  long_maps_kernel with limit = kLongPrefix: the state after the chunk's first 64 bytes, from every state;
//   B  long_continue_kernel: lane = (chunk, slot j < kLongSlots): the j-th DISTINCT state among those D, stepped through
//      the rest of the chunk - 4 lanes per chunk instead of D; a chunk with more distinct states is flagged;
//   A' long_maps_kernel with limit = chunk for the flagged chunks only (the old way: automata that count, a{1,200} on a's);
//   C  long_expand_kernel: map[s] = result of the slot that holds prefix_state[s].
constexpr uint32_t kLongPrefix = 64, kLongSlots = 4;
__device__ __forceinline__ void long_load_wide(const DfaDevice &p, uint16_t *wide) {
    const uint32_t D = p.nstates;
    for (uint32_t i = threadIdx.x; i < D * kWideColumns; i += blockDim.x) {
        const uint32_t s = i / kWideColumns, c = i % kWideColumns;
        wide[i] = (uint16_t)(p.next[s * p.ncls + p.cls[c]] * kWideColumns);     // column 128 stands for every byte >= 0x80
    }
}
// the row after bytes [pos, b) from `row` (pos 16-byte aligned if the base pointer is)
__device__ __forceinline__ uint32_t long_walk(const uint16_t *wide, const uint8_t *__restrict__ bytes, size_t pos, size_t b, uint32_t row) {
    if ((reinterpret_cast<uintptr_t>(bytes + pos) & 15) == 0) {
        for (; pos + 16 <= b; pos += 16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(bytes + pos);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t c = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
                row = wide[row + (c < 128 ? c : 128)];
            }
        }
    }
    for (; pos < b; pos++) {
        const uint32_t c = bytes[pos];
        row = wide[row + (c < 128 ? c : 128)];
    }
    return row;
}
// maps[k][s] = state after the first `limit` bytes of chunk k from state s; with `only`: just the chunks flagged there
__global__ __launch_bounds__(kLongThreads) void long_maps_kernel(DfaDevice p, const uint8_t *__restrict__ bytes, size_t nbytes, uint32_t chunk,
                                                                 uint32_t nchunks, uint16_t *__restrict__ maps, uint32_t limit,
                                                                 const uint8_t *__restrict__ only) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *wide = reinterpret_cast<uint16_t *>(smem);
    const uint32_t D = p.nstates;
    const uint32_t per_block = kLongThreads / D, ci = threadIdx.x / D, s0 = threadIdx.x % D;
    // the workgroups take batch after batch of per_block chunks: the table (D * 129 entries, two dependent loads and a
    // division each) is built once per workgroup, not once per batch
    bool loaded = false;
    for (size_t k0 = (size_t)blockIdx.x * per_block; k0 < nchunks; k0 += (size_t)gridDim.x * per_block) {
        const size_t k = k0 + ci;
        const bool mine = ci < per_block && k < nchunks && (!only || only[k]);
        if (only && !__syncthreads_or(mine ? 1 : 0)) continue;       // nothing flagged in this batch
        if (!loaded) { long_load_wide(p, wide); __syncthreads(); loaded = true; }
        if (!mine) continue;
        const size_t a = k * (size_t)chunk;
        size_t b = a + chunk < nbytes ? a + chunk : nbytes;
        if (a + limit < b) b = a + limit;
        maps[k * D + s0] = (uint16_t)(long_walk(wide, bytes, a, b, s0 * kWideColumns) / kWideColumns);
    }
}
// B: lane = (chunk, slot).  pre[k][*] = the D prefix states of chunk k (step A).  dist[k][j] = j-th distinct one (0xffff: none),
// res[k][j] = the state it reaches at the end of the chunk; flags[k] = 1 if there are more than kLongSlots.
__global__ __launch_bounds__(kLongThreads) void long_continue_kernel(DfaDevice p, const uint8_t *__restrict__ bytes, size_t nbytes, uint32_t chunk,
                                                                     uint32_t nchunks, const uint16_t *__restrict__ pre, uint16_t *__restrict__ dist,
                                                                     uint16_t *__restrict__ res, uint8_t *__restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *wide = reinterpret_cast<uint16_t *>(smem);
    const uint32_t D = p.nstates;
    long_load_wide(p, wide);
    __syncthreads();
    const size_t lane = (size_t)blockIdx.x * kLongThreads + threadIdx.x;
    const size_t k = lane / kLongSlots;
    const uint32_t j = (uint32_t)(lane % kLongSlots);
    if (k >= nchunks) return;
    uint32_t list[kLongSlots];
    uint32_t cnt = 0;                                                // distinct prefix states seen, in order of first appearance
#pragma unroll
    for (uint32_t i = 0; i < kLongSlots; i++) list[i] = 0xffffu;
    for (uint32_t s = 0; s < D && cnt <= kLongSlots; s++) {
        const uint32_t v = pre[k * D + s];
        bool seen = false;
#pragma unroll
        for (uint32_t i = 0; i < kLongSlots; i++) seen |= i < cnt && list[i] == v;
        if (!seen) {
#pragma unroll
            for (uint32_t i = 0; i < kLongSlots; i++)
                if (i == cnt) list[i] = v;
            cnt++;
        }
    }
    if (cnt > kLongSlots) { if (j == 0) flags[k] = 1; return; }
    uint32_t mine = 0xffffu;
#pragma unroll
    for (uint32_t i = 0; i < kLongSlots; i++)
        if (i == j) mine = list[i];
    dist[k * kLongSlots + j] = (uint16_t)mine;
    if (j == 0) flags[k] = 0;
    if (mine == 0xffffu) return;
    const size_t a = k * (size_t)chunk, b = a + chunk < nbytes ? a + chunk : nbytes;
    const size_t from = a + kLongPrefix < b ? a + kLongPrefix : b;
    res[k * kLongSlots + j] = (uint16_t)(long_walk(wide, bytes, from, b, mine * kWideColumns) / kWideColumns);
}
// C: lane = (chunk, state), in place: maps[k][s] holds the prefix state and receives the chunk's map entry
__global__ __launch_bounds__(kLongThreads) void long_expand_kernel(uint16_t *__restrict__ maps, uint32_t D, uint32_t nchunks,
                                                                   const uint16_t *__restrict__ dist, const uint16_t *__restrict__ res,
                                                                   const uint8_t *__restrict__ flags) {
    const size_t i = (size_t)blockIdx.x * kLongThreads + threadIdx.x;
    if (i >= (size_t)nchunks * D) return;
    const size_t k = i / D;
    if (flags[k]) return;                                            // built the old way (step A')
    const uint32_t v = maps[i];
    uint32_t out = 0;
#pragma unroll
    for (uint32_t j = 0; j < kLongSlots; j++)
        if (dist[k * kLongSlots + j] == v) out = res[k * kLongSlots + j];
    maps[i] = (uint16_t)out;
}
// out[g] = in[g*group + group-1] o ... o in[g*group]   (one lane per start state).  The group's maps are copied into LDS
// first (coalesced) and composed from there: composing straight from HBM/L2 was `group` dependent round trips per level,
// 90 us for 128, and three levels were most of the time of a string of a few MiB.
__global__ __launch_bounds__(kLongThreads) void long_compose_kernel(const uint16_t *__restrict__ in, uint32_t nin, uint32_t D, uint32_t group,
                                                                    uint16_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *sm = reinterpret_cast<uint16_t *>(smem);
    const size_t lo = (size_t)blockIdx.x * group, hi = lo + group < nin ? lo + group : nin;
    const uint32_t cnt = (uint32_t)(hi - lo);
    for (uint32_t i = threadIdx.x; i < cnt * D; i += kLongThreads) sm[i] = in[lo * D + i];
    __syncthreads();
    const uint32_t j = threadIdx.x;
    if (j >= D) return;
    uint32_t s = j;
    for (uint32_t k = 0; k < cnt; k++) s = sm[k * D + s];
    out[(size_t)blockIdx.x * D + j] = (uint16_t)s;
}
__global__ void long_finish_kernel(const uint16_t *__restrict__ map, DfaDevice p, uint8_t *__restrict__ accept) {
    if (threadIdx.x == 0 && blockIdx.x == 0) accept[0] = p.acc[map[p.start]];
}

}  // namespace

int count_newlines_per_stripe(const uint8_t *bytes, size_t nbytes, uint32_t stripe, uint32_t *counts, size_t nstripes, uint32_t *flags,
                              void *stream) {
    if (!nstripes) return 0;
    // (from 4 KiB stripes on: 8 GiB 1.63 ms against 1.8-1.95; at 1 KiB stripes the reduction per stripe makes it the slower
    // of the two, 2.4 ms against 1.7)
    if (stripe % 1024 == 0 && stripe >= 4096) {                      // a wave per stripe, the waves take stripe after stripe
        size_t blocks = (nstripes + 3) / 4;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(count_newlines_wave_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, bytes, nbytes, stripe, counts, nstripes, flags);
        return (int)hipGetLastError();
    }
    size_t blocks = (nstripes + 255) / 256;
    hipLaunchKernelGGL(count_newlines_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, bytes, nbytes, stripe, counts, nstripes, flags);
    return (int)hipGetLastError();
}
int expand_bits(const uint32_t *bits, size_t nlines, uint8_t *out, void *stream) {
    if (!nlines) return 0;
    size_t words = (nlines + 31) / 32, blocks = (words + 255) / 256;
    hipLaunchKernelGGL(expand_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, bits, nlines, out);
    return (int)hipGetLastError();
}
int scan_counts(const uint32_t *counts, uint64_t *base, uint64_t *chunk_sums, size_t n, void *stream) {
    const size_t nchunks = (n + kScanChunk - 1) / kScanChunk;
    hipStream_t st = (hipStream_t)stream;
    if (nchunks) hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3((unsigned)nchunks), dim3(256), 0, st, counts, n, chunk_sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, chunk_sums, nchunks, base + n);
    if (nchunks) hipLaunchKernelGGL(scan_chunks_kernel, dim3((unsigned)nchunks), dim3(256), 0, st, counts, n, chunk_sums, base);
    return (int)hipGetLastError();
}
size_t scan_scratch_words(size_t n) { return (n + kScanChunk - 1) / kScanChunk + 1; }

int match_stripes_dfa(const LineDfaDevice &p, bool clamp_high, const uint8_t *bytes, size_t nbytes, uint32_t stripe,
                      const uint64_t *stripe_base, size_t nstripes, uint32_t *accept, void *stream) {
#define GO(WIDE, CLAMP) launch_stripes<LineDfaEngine<WIDE, CLAMP>, LineDfaDevice>(p, LineDfaEngine<WIDE, CLAMP>::lds_bytes(p), bytes, nbytes, stripe, stripe_base, nstripes, accept, stream)
    if (p.in_global) return launch_stripes<LineDfaGlobalEngine, LineDfaDevice>(p, 256, bytes, nbytes, stripe, stripe_base, nstripes, accept, stream);
    if (p.wide) return clamp_high ? GO(true, true) : GO(true, false);
    return GO(false, false);
#undef GO
}
// slots (16 bytes of a lane's text) between two flushes of ALL lanes, minus one: about sixteen line ends per period
uint32_t flush_mask_for(size_t nbytes, size_t nlines) {
    const size_t avg = nlines ? nbytes / nlines : nbytes;
    uint32_t slots = 1;
    while (slots < 32 && (size_t)slots * 2 * 16 <= avg * 16) slots *= 2;      // (measured: profiles/r04_flush_period_ab.txt)
    return slots - 1;
}
int match_stripes_dfa2(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                       size_t nstripes, uint32_t *accept, void *stream, uint32_t flush_mask) {
    if (!nstripes) return 0;
    if (Dfa2::lds_bytes(p) > kDfa2MaxTable) return (int)hipErrorInvalidValue;
    size_t blocks = (nstripes + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(match_stripes2_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, p, bytes, nbytes, stripe, stripe_base, accept, flush_mask);
    return (int)hipGetLastError();
}
// ---- the sampled-table engine's second step: the two-bit bitmap (bit 2i = line i accepted, bit 2i + 1 = line i ended in the
// ESCAPE state) taken apart into the accept bitmap - every word written, nothing to clear beforehand - and the bitmap of the
// escaped lines, which the exact engine then decides (recheck_escaped: kernels_nfa.inc); their number is added up on the side.
// `list` receives the numbers of the escaped lines (in no order; `cap` entries: what does not fit is counted all the same, and
// recheck_escaped then walks the stripes instead).  A workgroup collects its lines in LDS and reserves room in the list a
// thousand at a time: one atomic per escaped line - or per wave that met one - on the one counter cost a millisecond at
// 400 000 escaped lines.
__global__ __launch_bounds__(256) void split_two_bit_kernel(const uint32_t *__restrict__ wide, size_t words, uint32_t *__restrict__ accept_bits,
                                                             uint32_t *__restrict__ escaped_bits, unsigned long long *__restrict__ escaped_total,
                                                             uint64_t *__restrict__ list, size_t cap) {
    constexpr uint32_t kLocal = 4096, kFlushAt = kLocal - 256 * 8;      // (a turn adds at most 256 x 32 lines: see the two-step append below)
    __shared__ uint64_t local[kLocal];
    __shared__ uint32_t nlocal;
    __shared__ unsigned long long flush_base;
    if (threadIdx.x == 0) nlocal = 0;
    __syncthreads();
    auto even_bits = [](uint32_t x) -> uint32_t {                   // bits 0, 2, 4, ... packed into the low half
        x &= 0x55555555u;
        x = (x | x >> 1) & 0x33333333u;
        x = (x | x >> 2) & 0x0f0f0f0fu;
        x = (x | x >> 4) & 0x00ff00ffu;
        return (x | x >> 8) & 0xffffu;
    };
    auto flush = [&]() {                                             // whole workgroup; nlocal is stable here
        __syncthreads();
        const uint32_t n = nlocal;
        if (n) {
            if (threadIdx.x == 0) flush_base = atomicAdd(escaped_total, (unsigned long long)n);
            __syncthreads();
            const unsigned long long base = flush_base;
            for (uint32_t i = threadIdx.x; i < n; i += 256) if (base + i < cap) list[base + i] = local[i];
            __syncthreads();
            if (threadIdx.x == 0) nlocal = 0;
        }
        __syncthreads();
    };
    const size_t turns = (words + (size_t)gridDim.x * 256 - 1) / ((size_t)gridDim.x * 256);       // the same for every lane: the flushes are collective
    for (size_t it = 0; it < turns; it++) {
        const size_t w = (it * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        uint32_t esc = 0;
        if (w < words) {
            const uint2 v = reinterpret_cast<const uint2 *>(wide)[w];
            accept_bits[w] = even_bits(v.x) | even_bits(v.y) << 16;
            esc = even_bits(v.x >> 1) | even_bits(v.y >> 1) << 16;
            escaped_bits[w] = esc;
        }
        // append my escaped lines, eight at a time (so that a turn never overruns the local buffer between two flushes)
        while (__syncthreads_or(esc != 0)) {
            uint32_t take = 0;
            for (uint32_t k = 0, m = esc; k < 8 && m; k++) { take |= m & (0u - m); m &= m - 1; }
            const uint32_t c = (uint32_t)__popc(take);
            if (c) {
                uint32_t at = atomicAdd(&nlocal, c);
                for (uint32_t m = take; m; m &= m - 1) local[at++] = (uint64_t)w * 32 + (uint32_t)(__ffs((int)m) - 1);
            }
            esc &= ~take;
            __syncthreads();
            if (nlocal > kFlushAt) flush();
        }
    }
    flush();
}
int match_stripes_dfa2_two_bit(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                               size_t nstripes, uint32_t *wide_bits, void *stream) {
    if (!nstripes) return 0;
    if (Dfa2::lds_bytes(p) > kDfa2MaxTable) return (int)hipErrorInvalidValue;
    size_t blocks = (nstripes + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(match_stripes2_two_bit_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, p, bytes, nbytes, stripe, stripe_base, wide_bits);
    return (int)hipGetLastError();
}
int split_two_bit(const uint32_t *wide, size_t nlines, uint32_t *accept_bits, uint32_t *escaped_bits, unsigned long long *escaped_total, uint64_t *list,
                  size_t cap, void *stream) {
    const size_t words = (nlines + 31) / 32;
    if (!words) return 0;
    size_t blocks = (words + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(split_two_bit_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, wide, words, accept_bits, escaped_bits, escaped_total,
                       list, cap);
    return (int)hipGetLastError();
}
int match_units_dfa2(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                     size_t nstripes, uint32_t *accept, uint32_t units_per_wg, void *stream) {
    if (!nstripes) return 0;
    if (Dfa2::lds_bytes(p) > kDfa2MaxTable || !units_per_wg) return (int)hipErrorInvalidValue;
    const size_t units = (nstripes + 63) / 64, blocks = (units + units_per_wg - 1) / units_per_wg;
    hipLaunchKernelGGL(match_units2_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, p, bytes, nbytes, stripe, stripe_base, accept, units_per_wg);
    return (int)hipGetLastError();
}
// byte-stride table engines in one-pass mode (bytes >= 0x80 are always clamped: nobody has looked at the corpus yet)
int match_onepass_dfa(const LineDfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, size_t nstripes, uint32_t *counts,
                      uint32_t *slabs, void *stream) {
    if (p.in_global) return launch_onepass<LineDfaGlobalEngine, LineDfaDevice>(p, 256, bytes, nbytes, stripe, nstripes, counts, slabs, stream);
    if (p.wide) return launch_onepass<LineDfaEngine<true, true>, LineDfaDevice>(p, LineDfaEngine<true, true>::lds_bytes(p), bytes, nbytes, stripe, nstripes, counts, slabs, stream);
    return launch_onepass<LineDfaEngine<false, false>, LineDfaDevice>(p, LineDfaEngine<false, false>::lds_bytes(p), bytes, nbytes, stripe, nstripes, counts, slabs, stream);
}
size_t onepass_slab_words(size_t nstripes, uint32_t stripe) {
    return ((nstripes + kThreads - 1) / kThreads) * slab_words_per_lane(stripe) * kThreads;
}
int match_onepass_dfa2(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, size_t nstripes, uint32_t *counts,
                       uint32_t *slabs, void *stream) {
    if (!nstripes) return 0;
    if (Dfa2::lds_bytes(p) > kDfa2MaxTable) return (int)hipErrorInvalidValue;
    size_t blocks = (nstripes + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(match_stripes2_onepass_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, p, bytes, nbytes, stripe, counts, slabs);
    return (int)hipGetLastError();
}
int compact_streams(const uint32_t *counts, const uint64_t *stripe_base, size_t nstripes, uint32_t stripe, const uint32_t *slabs,
                    uint32_t *accept_bits, size_t cap_words, void *stream) {
    if (!nstripes) return 0;
    hipLaunchKernelGGL(compact_streams_kernel, dim3((unsigned)((nstripes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, counts, stripe_base, nstripes,
                       stripe, slabs, accept_bits, cap_words);
    return (int)hipGetLastError();
}
int mail_results(const uint64_t *total, const uint32_t *flags, const uint8_t *last_byte, uint64_t *mail, void *stream) {
    hipLaunchKernelGGL(mail_results_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, total, flags, last_byte, mail);
    return (int)hipGetLastError();
}
int build_line_offsets(const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base, size_t nstripes,
                       uint64_t *line_off, void *stream) {
    if (!nstripes) return 0;
    hipLaunchKernelGGL(line_offsets_kernel, dim3((unsigned)((nstripes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bytes, nbytes, stripe,
                       stripe_base, nstripes, line_off);
    return (int)hipGetLastError();
}
int empty_matches(const uint64_t *line_off, size_t nlines, uint32_t *count, const uint64_t *first, uint32_t *match_start, uint32_t *match_end,
                  void *stream, size_t cap) {
    if (!nlines) return 0;
    const dim3 grid((unsigned)((nlines + 255) / 256));
    if (first) hipLaunchKernelGGL(empty_matches_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, line_off, nlines, count, first, match_start, match_end, (uint64_t)cap);
    else hipLaunchKernelGGL(empty_matches_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, line_off, nlines, count, first, match_start, match_end, (uint64_t)cap);
    return (int)hipGetLastError();
}
static uint32_t long_chunk(size_t nbytes) {
    // short strings: 256-byte chunks (a string of a few KiB is a handful of short launches, not one long sequential lane);
    // from 256 KiB on chunks of 1 KiB or more, at most 65536 of them
    uint32_t chunk = 256;
    while (((nbytes + chunk - 1) / chunk) > (chunk < 1024 ? 1024u : 65536u)) chunk <<= 1;
    return chunk;
}
size_t long_scratch_bytes(uint32_t nstates, size_t nbytes, uint32_t *chunk) {
    *chunk = long_chunk(nbytes);
    const size_t k0 = (nbytes + *chunk - 1) / *chunk, k1 = (k0 + kLongGroup - 1) / kLongGroup;
    // level 0, then the levels ping-pong between two areas; then per chunk kLongSlots distinct prefix states and their
    // results (u16 each) and a flag byte
    return (k0 + k1 + 2) * nstates * sizeof(uint16_t) + k0 * (kLongSlots * 2 * sizeof(uint16_t) + 1) + 16;
}
int match_long_dfa(const DfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t chunk, void *scratch, uint8_t *accept,
                   void *stream) {
    const uint32_t D = p.nstates;
    if (!D || D > kLongMaxStates || !nbytes) return (int)hipErrorInvalidValue;
    const size_t lds = (size_t)D * kWideColumns * sizeof(uint16_t);
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(long_maps_kernel), lds);
    if (e != hipSuccess) return (int)e;
    static LdsAttr attr2, attr3;
    e = ensure_dynamic_lds(attr2, reinterpret_cast<const void *>(long_continue_kernel), lds);
    if (e == hipSuccess) e = ensure_dynamic_lds(attr3, reinterpret_cast<const void *>(long_compose_kernel), (size_t)kLongGroup * D * sizeof(uint16_t));
    if (e != hipSuccess) return (int)e;
    uint32_t n = (uint32_t)((nbytes + chunk - 1) / chunk);
    uint16_t *cur = static_cast<uint16_t *>(scratch), *other = cur + (size_t)n * D;
    const size_t k1 = ((size_t)n + kLongGroup - 1) / kLongGroup;
    uint16_t *dist = cur + ((size_t)n + k1 + 2) * D, *res = dist + (size_t)n * kLongSlots;
    uint8_t *flags = reinterpret_cast<uint8_t *>(res + (size_t)n * kLongSlots);
    const uint32_t per_block = kLongThreads / D;
    const uint32_t batches = (n + per_block - 1) / per_block;
    const dim3 by_state(batches < 2048 ? batches : 2048);
    hipStream_t st = (hipStream_t)stream;
    if (chunk > 2 * kLongPrefix) {
        hipLaunchKernelGGL(long_maps_kernel, by_state, dim3(kLongThreads), lds, st, p, bytes, nbytes, chunk, n, cur, kLongPrefix, nullptr);
        hipLaunchKernelGGL(long_continue_kernel, dim3((unsigned)(((size_t)n * kLongSlots + kLongThreads - 1) / kLongThreads)), dim3(kLongThreads), lds, st, p,
                           bytes, nbytes, chunk, n, cur, dist, res, flags);
        hipLaunchKernelGGL(long_maps_kernel, by_state, dim3(kLongThreads), lds, st, p, bytes, nbytes, chunk, n, cur, chunk, flags);
        hipLaunchKernelGGL(long_expand_kernel, dim3((unsigned)(((size_t)n * D + kLongThreads - 1) / kLongThreads)), dim3(kLongThreads), 0, st, cur, D, n, dist, res,
                           flags);
    } else {
        hipLaunchKernelGGL(long_maps_kernel, by_state, dim3(kLongThreads), lds, st, p, bytes, nbytes, chunk, n, cur, chunk, nullptr);
    }
    uint16_t *area[2] = {other, cur};                          // level 1 writes behind level 0, level 2 over level 0, ...
    for (int lvl = 0; n > 1; lvl++) {
        const uint32_t m = (n + kLongGroup - 1) / kLongGroup;
        uint16_t *dst = area[lvl & 1];
        hipLaunchKernelGGL(long_compose_kernel, dim3(m), dim3(kLongThreads), (size_t)kLongGroup * D * sizeof(uint16_t), (hipStream_t)stream, cur, n, D, kLongGroup, dst);
        cur = dst;
        n = m;
    }
    hipLaunchKernelGGL(long_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cur, p, accept);
    return (int)hipGetLastError();
}
// The index of a batch of items (kept by an rrx_items handle, or built in scratch by rrx_match_extents):
//   [ends bitmap, 1 bit per byte | flag u32 (an item without a byte for its mark) | stripe base u64 (nstripes + 1)]
// and, per match, a result bitmap of nitems bits.
static size_t items_align(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t items_ends_bytes(size_t nbytes, uint32_t stripe) {           // whole groups of 64 stripes (ends_slot permutes inside a group)
    const size_t group_words = 2 * (size_t)stripe, words = (nbytes + 31) / 32 + 4;
    return (words + group_words - 1) / group_words * group_words * 4;
}
// the stripe an items batch wants: by its size and its mean item length, like a corpus (stripe_for_lines)
static uint32_t items_stripe(size_t nbytes, size_t nitems) { return stripe_for_lines(nbytes, nitems ? nbytes / nitems : nbytes); }
size_t items_index_bytes(size_t nbytes, size_t nitems) {
    const size_t nstripes = (nbytes + items_stripe(nbytes, nitems) - 1) / items_stripe(nbytes, nitems);
    return items_ends_bytes(nbytes, items_stripe(nbytes, nitems)) + 256 + items_align((nstripes + 1) * 8);
}
size_t items_result_bytes(size_t nitems) { return items_align(((nitems + 31) / 32 + 4) * 4); }
// trim 0 or 1; the buffer starts at off[0] and holds nbytes = off[nitems] - off[0] bytes.  -> *flag: device u32 inside the
// index, != 0 after the stream is done if some item has no byte for its mark (then the index is not usable).
int items_index_build(size_t nbytes, const uint64_t *off, size_t nitems, uint32_t trim, void *index, uint32_t **flag, void *stream,
                      const uint8_t *resolve_base, size_t min_bytes) {
    if (trim > 1 || !nitems || !nbytes) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t stripe = items_stripe(nbytes, nitems);
    const size_t nstripes = (nbytes + stripe - 1) / stripe;
    uint32_t *ends = static_cast<uint32_t *>(index);
    uint32_t *fl = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(index) + items_ends_bytes(nbytes, stripe));
    uint64_t *base = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(fl) + 256);
    *flag = fl;
    if (stripe & (stripe - 1)) return (int)hipErrorInvalidValue;      // (stripes are powers of two)
    hipError_t e = hipMemsetAsync(fl, 0, 256, st);                    // the flag; the bitmap is written whole by the kernel
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(item_index_kernel, dim3((unsigned)((nitems + kEndsItems - 1) / kEndsItems)), dim3(256), 0, st, off, nitems, trim, ends, fl,
                       (uint64_t)(items_ends_bytes(nbytes, stripe) / 4), resolve_base, (uint64_t)min_bytes, (uint32_t)__builtin_ctz(stripe), nstripes, base);
    return (int)hipGetLastError();
}
// one byte per item into `accept` (16-byte aligned); `result` = items_result_bytes(nitems) of scratch.  resolve_off != nullptr:
// the one-call form - `bytes` is the buffer the offsets index, `nbytes` the upper bound the index was laid out for, the
// kernel takes the batch's start and length from the offsets and does nothing if *skip_if != 0.
int items_match2(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, size_t nitems, const void *index, void *result, uint8_t *accept, void *stream,
                 const uint64_t *resolve_off, const uint32_t *skip_if) {
    if (!p.P || !p.T2 || Dfa2::lds_bytes(p) > kDfa2MaxTable || !nitems || !nbytes) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t stripe = items_stripe(nbytes, nitems);
    const size_t nstripes = (nbytes + stripe - 1) / stripe;
    const uint32_t *ends = static_cast<const uint32_t *>(index);
    const uint64_t *base = reinterpret_cast<const uint64_t *>(static_cast<const uint8_t *>(index) + items_ends_bytes(nbytes, stripe) + 256);
    uint32_t *bits = static_cast<uint32_t *>(result);
    hipError_t e = hipMemsetAsync(bits, 0, ((nitems + 31) / 32 + 4) * 4, st);
    if (e != hipSuccess) return (int)e;
    const size_t blocks = (nstripes + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(match_items_stripes2_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, p, bytes, nbytes, stripe, base, ends, bits, resolve_off, nitems, skip_if);
    const int rc = (int)hipGetLastError();
    if (rc) return rc;
    return expand_bits(bits, nitems, accept, stream);
}
int items_match(const LineDfaDevice &p, const uint8_t *bytes, size_t nbytes, size_t nitems, uint32_t trim, const void *index, void *result,
                uint8_t *accept, void *stream, const uint64_t *resolve_off, const uint32_t *skip_if) {
    if (!p.wide || p.in_global || p.stride != (kItemColumns << p.rep_log2) || trim > 1 || !nitems || !nbytes) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t stripe = items_stripe(nbytes, nitems);
    const size_t nstripes = (nbytes + stripe - 1) / stripe;
    const uint32_t *ends = static_cast<const uint32_t *>(index);
    const uint64_t *base = reinterpret_cast<const uint64_t *>(static_cast<const uint8_t *>(index) + items_ends_bytes(nbytes, stripe) + 256);
    uint32_t *bits = static_cast<uint32_t *>(result);
    hipError_t e = hipMemsetAsync(bits, 0, ((nitems + 31) / 32 + 4) * 4, st);
    if (e != hipSuccess) return (int)e;
    const size_t table_bytes = LineDfaEngine<true, false>::lds_bytes(p);
    const uint32_t stage_off = (uint32_t)((table_bytes + 15) & ~(size_t)15);
    const size_t half_cu = 80 * 1024;
    const uint32_t stage_words = stage_off + kStageWords * sizeof(uint32_t) >= half_cu ? kStageWords : (uint32_t)((half_cu - stage_off) / 4);
    const size_t lds = stage_off + (size_t)stage_words * sizeof(uint32_t);
    const size_t blocks = (nstripes + kThreads - 1) / kThreads;
    if (trim == 1) {
        static LdsAttr attr;
        e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(match_items_stripes_kernel<1>), lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(match_items_stripes_kernel<1>, dim3((unsigned)blocks), dim3(kThreads), lds, st, p, bytes, nbytes, stripe, base, ends, bits, stage_off, stage_words,
                           resolve_off, nitems, skip_if);
    } else {
        static LdsAttr attr;
        e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(match_items_stripes_kernel<2>), lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(match_items_stripes_kernel<2>, dim3((unsigned)blocks), dim3(kThreads), lds, st, p, bytes, nbytes, stripe, base, ends, bits, stage_off, stage_words,
                           resolve_off, nitems, skip_if);
    }
    const int rc = (int)hipGetLastError();
    if (rc) return rc;
    return expand_bits(bits, nitems, accept, stream);
}
int match_extents_dfa(const DfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                      void *stream, const uint32_t *only_if) {
    // tables beyond the LDS budget (the batch kernel's "global" form) stay in HBM/L2 here too
    if (PlainDfaEngine::lds_bytes(p) > kPlainDfaLdsBudget)
        return launch_extents<PlainDfaGlobalEngine, DfaDevice>(p, PlainDfaGlobalEngine::lds_bytes(p), bytes, off, nitems, trim, accept, stream, only_if);
    return launch_extents<PlainDfaEngine, DfaDevice>(p, PlainDfaEngine::lds_bytes(p), bytes, off, nitems, trim, accept, stream, only_if);
}

}  // namespace dev
}  // namespace rrx
