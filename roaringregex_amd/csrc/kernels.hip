// kernels.hip — hand-written CDNA4 (gfx950) kernels for the RoaringRegex hot path.
//
// Replaces, for a whole batch of '\n'-delimited strings at once:
//   AcceptanceIterator::operator++(int)   regex.h:156-159   (consume the string)
//   Processor::shift<true>                NFA.cc:72-102     (per-byte state-set transition)
//   Processor::operator*()                NFA.cc:103-107    (accepting?)
// Pure integer/bitwise work, HBM-read bound by design: no MFMA.
//
// Batch kernel (match_stripes<Engine>) — the text never touches LDS:
//   * lane g of the grid owns the lines that START in its contiguous stripe (1-16 KiB) of the corpus and follows
//     its last line past the stripe end, so every line is stepped by exactly one lane from its first byte;
//   * each lane streams its own stripe from HBM straight into registers, one whole 128-byte line
//     (8 x global_load_dwordx4) per round through a rotating 8-slot register buffer: slot i is refilled for
//     the next round right after it has been consumed, so every fetched line is used up while it is still
//     resident (a 64-byte round re-fetched the second half of every line: measured 1.73x read traffic);
//   * the automaton tables live in LDS, the state lives in registers;
//   * line verdicts are accumulated in registers as ordered bits, packed into the lane's current 32-bit
//     output word and merged into the accept BITMAP (bit i = line i) with one global atomic OR per filled
//     word (about one per 32 lines; per-line byte stores cost 23x their size in HBM write traffic);
//     line index = stripe_base[g] + newlines seen so far, stripe_base being the per-stripe newline prefix
//     the corpus carries (8 bytes per stripe of text): no per-line offset array is ever read.
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

#include "device.hpp"


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
        for (int i = threadIdx.x; i < n; i += blockDim.x) t[i] = p.table[i] + base;
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

// ---- shift-and NFA: state set in W registers --------------------------------------------------------
template <int W>
struct NfaCore {
    struct State { uint32_t s[W]; };
    const uint32_t *B;      // LDS [256][W]
    const uint32_t *X;      // LDS [nbits][W]
    NfaMasks m;
    bool any_exc, any_carry;

    static size_t lds_bytes(const NfaDevice &p) { return ((size_t)256 * W + (size_t)p.nbits * W) * 4; }
    __device__ void load(const NfaDevice &p, uint8_t *lds) {
        uint32_t *b = reinterpret_cast<uint32_t *>(lds);
        uint32_t *x = b + 256 * W;
        for (int i = threadIdx.x; i < 256 * W; i += blockDim.x) b[i] = p.B[i];
        for (int i = threadIdx.x; i < (int)p.nbits * W; i += blockDim.x) x[i] = p.X[i];
        B = b; X = x; m = p.masks; any_exc = p.any_exc != 0; any_carry = p.any_carry != 0;
    }
    __device__ __forceinline__ bool accepting(const State &st) const {
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < W; i++) a |= st.s[i] & m.fin[i];
        return a != 0;
    }
    // next = ( ((S << 1) & CHAIN) | (S & SELF) | (((S & CGRP) + CGRP) & CTGT) | OR_{e in S & EXC} X[e] ) & B[c]
    __device__ __forceinline__ void advance(State &st, uint32_t c) const {
        uint32_t t[W];
        uint32_t exc = 0;
#pragma unroll
        for (int i = 0; i < W; i++) {
            uint32_t lo = i ? st.s[i - 1] : 0u;
            uint32_t sh = __builtin_amdgcn_alignbit(st.s[i], lo, 31);
            t[i] = (sh & m.chain[i]) | (st.s[i] & m.self[i]);
            exc |= st.s[i] & m.excm[i];
        }
        if (any_carry) {                                     // ((S & CGRP) + CGRP) & CTGT as one multiword add
            uint32_t cy = 0;
#pragma unroll
            for (int i = 0; i < W; i++) {
                const uint64_t sum = (uint64_t)(st.s[i] & m.cgrp[i]) + m.cgrp[i] + cy;
                t[i] |= (uint32_t)sum & m.ctgt[i];
                cy = (uint32_t)(sum >> 32);
            }
        }
        if (any_exc && exc) {
#pragma unroll
            for (int i = 0; i < W; i++) {
                uint32_t e = st.s[i] & m.excm[i];
                while (e) {
                    int b = __ffs(e) - 1;
                    e &= e - 1;
                    const uint32_t *row = X + (size_t)(32 * i + b) * W;
#pragma unroll
                    for (int j = 0; j < W; j++) t[j] |= row[j];
                }
            }
        }
        const uint32_t *bc = B + c * W;
#pragma unroll
        for (int i = 0; i < W; i++) st.s[i] = t[i] & bc[i];
    }
};

// ---- line-mode shift-and NFA (the batch kernel's NFA engine): NFA.cc:86-100 restated for one string per lane with the
// state set in W registers and the per-character bitmaps B[c] in LDS.  Everything the line protocol needs is folded
// into the automaton, so that a byte costs no compare and no re-initialisation:
//   * position 0 is the initial state.  The shift injects a 1 into it on EVERY byte, t0 = (S0 << 1) | 1, and only
//     B['\n'] contains position 0: after a '\n' the set is exactly {initial}, after any other byte position 0 is
//     clear.  Bytes 0x00 and >= 0x80 have empty rows: the set dies and stays dead until the next '\n'.
//   * the lowering leaves a never-entered gap position in front of every path of the path cover (lower_nfa, gaps),
//     so the bit shifted out of a path's end dies in the gap: no CHAIN mask.
//   * the verdict (S & FIN != 0, NFA.cc:103-107) is only evaluated in byte steps where SOME lane of the wave sits on a
//     '\n' (one SDWA compare + a scalar branch otherwise).
// Per byte and word: 1 shift (v_alignbit / v_lshl_or) + 1 AND with the B word, + 1 v_and_or for the self loops when
// the automaton has any (SELF); CARRY adds the add-carry groups (3 per word); RULES adds the exception rows of NfaCore
// (a divergent loop over the live exception positions) and takes the carry groups under a run-time flag.
template <int W, bool SELF, bool RULES, bool CARRY = false>
struct LineNfaEngine : NfaCore<W> {
    // Measured on the builds with rules (profiles/r02_nfa_variants.txt): the LDS result window and the four-rows-ahead
    // fetch each cost them 4-20 % (registers; their time goes to the exception loop), the plain builds gain from both.
    static constexpr bool kStaged = !RULES;
    static constexpr bool kEightWaves = W <= 2 && !RULES;      // 64 registers = two workgroups per CU
    static constexpr int kRoundBytes = W <= 4 ? kRound : kRound / 2;
    using State = typename NfaCore<W>::State;
    __device__ void load(const NfaDevice &p, uint8_t *lds) {
        NfaCore<W>::load(p, lds);
        // The byte step uses byte * row bytes as the LDS address of a B row: B must sit at LDS address 0, which it does
        // as long as the kernel has no static LDS in front of its dynamic LDS.  (Adding the base would be a sixth VALU
        // instruction per byte: the base is a relocation, not an immediate.)
        if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds != 0u) __builtin_trap();
        __syncthreads();
        uint32_t *b = reinterpret_cast<uint32_t *>(lds);
        if (threadIdx.x < W) b[(size_t)'\n' * W + threadIdx.x] = threadIdx.x == 0 ? 1u : 0u;      // B['\n'] = {initial}
    }
    __device__ __forceinline__ State fresh() const {
        State st;
#pragma unroll
        for (int i = 0; i < W; i++) st.s[i] = i == 0 ? 1u : 0u;
        return st;
    }
    __device__ __forceinline__ State skipping() const {
        State st;
#pragma unroll
        for (int i = 0; i < W; i++) st.s[i] = 0;
        return st;
    }
    __device__ __forceinline__ void advance_line(State &st, uint32_t c) const { advance_row(st, this->B + c * W); }
    __device__ __forceinline__ void advance_row(State &st, const uint32_t *bc) const {
        uint32_t t[W];
        t[0] = (st.s[0] << 1) | 1u;
#pragma unroll
        for (int i = 1; i < W; i++) t[i] = __builtin_amdgcn_alignbit(st.s[i], st.s[i - 1], 31);
        if (SELF) {
#pragma unroll
            for (int i = 0; i < W; i++) t[i] |= st.s[i] & this->m.self[i];
        }
        if (RULES || CARRY) {
            if (CARRY || this->any_carry) {                      // ((S & CGRP) + CGRP) & CTGT as one multiword add
                uint32_t cy = 0;
#pragma unroll
                for (int i = 0; i < W; i++) {
                    const uint64_t sum = (uint64_t)(st.s[i] & this->m.cgrp[i]) + this->m.cgrp[i] + cy;
                    t[i] |= (uint32_t)sum & this->m.ctgt[i];
                    cy = (uint32_t)(sum >> 32);
                }
            }
            if (RULES && this->any_exc) {
                uint32_t exc = 0;
#pragma unroll
                for (int i = 0; i < W; i++) exc |= st.s[i] & this->m.excm[i];
                if (exc) {
#pragma unroll
                    for (int i = 0; i < W; i++) {
                        uint32_t e = st.s[i] & this->m.excm[i];
                        while (e) {
                            const int b = __ffs(e) - 1;
                            e &= e - 1;
                            const uint32_t *row = this->X + (size_t)(32 * i + b) * W;
#pragma unroll
                            for (int j = 0; j < W; j++) t[j] |= row[j];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < W; i++) st.s[i] = t[i] & bc[i];
    }
    // Byte K of text word w.  Two SDWA instructions read the byte straight out of the text register: the LDS address of
    // its B row (byte * row bytes; B is the first thing in LDS) and the lane mask "this byte is '\n'".  hipcc's own
    // code for the plain C form was 13 VALU per byte at W = 1 (field extraction, and the verdict if-converted into every
    // step).  The four B rows of a text word are requested before the first of the four dependent steps.
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
    struct Row { uint32_t b[W]; };
    template <int K>
    __device__ __forceinline__ Row row_of(uint32_t w) const {
        uint32_t off;
        const uint32_t row_bytes = W * 4;
#define RRX_NFA_ROW(SEL) asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" SEL " src1_sel:DWORD" : "=v"(off) : "v"(w), "v"(row_bytes))
        if constexpr (K == 0) RRX_NFA_ROW("BYTE_0");
        if constexpr (K == 1) RRX_NFA_ROW("BYTE_1");
        if constexpr (K == 2) RRX_NFA_ROW("BYTE_2");
        if constexpr (K == 3) RRX_NFA_ROW("BYTE_3");
#undef RRX_NFA_ROW
        const lds_u32_ptr p = reinterpret_cast<lds_u32_ptr>(off);    // B sits at LDS address 0 (checked in load)
        Row r;
#pragma unroll
        for (int i = 0; i < W; i++) r.b[i] = p[i];
        return r;
    }
    template <int K>
    __device__ __forceinline__ void verdict(const State &st, uint32_t w, uint32_t &bits) const {
        uint64_t nlmask;
        const uint32_t ten = '\n';
#define RRX_NFA_NL(SEL) asm("v_cmp_eq_u32_sdwa %0, %1, %2 src0_sel:" SEL " src1_sel:DWORD" : "=s"(nlmask) : "v"(w), "v"(ten))
        if constexpr (K == 0) RRX_NFA_NL("BYTE_0");
        if constexpr (K == 1) RRX_NFA_NL("BYTE_1");
        if constexpr (K == 2) RRX_NFA_NL("BYTE_2");
        if constexpr (K == 3) RRX_NFA_NL("BYTE_3");
#undef RRX_NFA_NL
        if (nlmask) {                                             // wave-uniform: some lane ends a line on this byte
            uint32_t a = 0, tmp;
#pragma unroll
            for (int i = 0; i < W; i++) a |= st.s[i] & this->m.fin[i];
            // bits = (bits << 1) | verdict on the lanes of nlmask: the verdict enters as the carry of bits + bits
            asm volatile("v_cmp_ne_u32_e32 vcc, 0, %2\n\t"
                         "v_addc_co_u32_e32 %1, vcc, %0, %0, vcc\n\t"
                         "v_cndmask_b32_e64 %0, %0, %1, %3"
                         : "+v"(bits), "=&v"(tmp) : "v"(a), "s"(nlmask) : "vcc");
        }
    }
    __device__ __forceinline__ void consume_word(State &st, uint32_t w, uint32_t &bits) const {
        if constexpr (W <= 4 && !RULES) {
            const Row r0 = row_of<0>(w), r1 = row_of<1>(w), r2 = row_of<2>(w), r3 = row_of<3>(w);
            verdict<0>(st, w, bits); advance_row(st, r0.b);
            verdict<1>(st, w, bits); advance_row(st, r1.b);
            verdict<2>(st, w, bits); advance_row(st, r2.b);
            verdict<3>(st, w, bits); advance_row(st, r3.b);
        } else {                                                  // wide sets: one row in flight ahead of the step
            Row r = row_of<0>(w), n = row_of<1>(w);
            verdict<0>(st, w, bits); advance_row(st, r.b);
            r = row_of<2>(w);
            verdict<1>(st, w, bits); advance_row(st, n.b);
            n = row_of<3>(w);
            verdict<2>(st, w, bits); advance_row(st, r.b);
            verdict<3>(st, w, bits); advance_row(st, n.b);
        }
    }
    __device__ __forceinline__ void step(State &st, uint32_t c, uint32_t &nl, uint32_t &acc) const {
        const bool isnl = c == '\n';
        nl = isnl ? 1u : 0u;
        acc = (isnl && this->accepting(st)) ? 1u : 0u;
        advance_line(st, c);
    }
};

// ---- plain engines for the extents kernel ('\n' is an ordinary byte) --------------------------------
template <int W>
struct PlainNfaEngine : NfaCore<W> {
    using State = typename NfaCore<W>::State;
    __device__ __forceinline__ void reset(State &st) const {
#pragma unroll
        for (int i = 0; i < W; i++) st.s[i] = this->m.init[i];
    }
    __device__ __forceinline__ void kill(State &st) const {
#pragma unroll
        for (int i = 0; i < W; i++) st.s[i] = 0;
    }
    __device__ __forceinline__ void step(State &st, uint32_t c) const { this->advance(st, c); }
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

constexpr uint32_t kEndsOnNewline = 0x80000000u;         // bit 31 of counts[k]: stripe k ends on a '\n'
constexpr uint64_t kFreshStripe = 1ull << 63;             // bit 63 of stripe_base[k]: stripe k begins at the start of a line
__device__ __forceinline__ uint64_t line_of(uint64_t base) { return base & ~kFreshStripe; }

// Text loads are plain loads: non-temporal ones stop the 8 loads of a 128-byte line from merging into one request
// (measured -47 %, profiles/r01_v7_result_path_probes.txt).
__device__ __forceinline__ uint4 load_text(const uint4 *p) { return *p; }

// One round of a lane's text: N 16-byte slots requested as ONE burst (they merge into one request per 128-byte line)
// and consumed in order.  The slots are members reached through compile-time recursion, never an indexed array: an
// engine whose step contains a loop keeps the compiler from unrolling a slot loop, and an indexed buffer then lives in
// scratch memory (measured on the first NFA engine: 144 bytes of scratch per lane, 18 ms per GiB).
template <int N>
struct TextRound {
    uint4 head;
    TextRound<N - 1> rest;
    __device__ __forceinline__ void load(const uint4 *p) { head = load_text(p); rest.load(p + 1); }
    template <class F> __device__ __forceinline__ void for_each_slot(F &&f) const { f(head); rest.for_each_slot(f); }
};
template <>
struct TextRound<0> {
    __device__ __forceinline__ void load(const uint4 *) {}
    template <class F> __device__ __forceinline__ void for_each_slot(F &&) const {}
};

// ============================================================================================ batch kernel
// Line verdicts of one lane.  `bits` = sentinel 1 followed by one verdict bit per line finished since the
// last flush (oldest highest).  flush() appends them, oldest first, to the lane's current output word at bit
// position `fill` and ORs every completed word into the accept bitmap (bit i of the bitmap = line i).
// The first result of a lane that started inside somebody else's line belongs to that somebody (who reports
// it when it follows the line past its own stripe): its bit is skipped but its index is consumed.
// STAGED: completed words are ORed into a per-workgroup LDS window of the bitmap (`stage`, kStageWords words from the
// word that holds the workgroup's first line) and written out by the whole workgroup at the end, 256 contiguous bytes
// per wave instruction.  Scattered 4-byte global atomics leave L2 as partial-line DRAM writes: 5.6 M of them per
// launch on the URL config cost 4-6 % of the kernel (probes: atomics confined to 16 KiB of L2, no memory operation).
// Words beyond the window (a workgroup whose lines average < 32 bytes) still go to memory directly.
constexpr uint32_t kStageWords = 4096;
template <bool STAGED>
struct ResultsT {
    uint32_t bits = 1;
    uint32_t outw = 0;
    uint32_t fill;
    uint32_t seen = 0;
    uint64_t word;                         // STAGED: relative to the first word of the window
    bool drop_first;
    bool writer = true;                    // wave-cooperative kernels: every lane mirrors the bookkeeping, one writes
    uint32_t *__restrict__ out;            // STAGED: already advanced to the first word of the window
    uint32_t *stage = nullptr;
    uint32_t stage_words = kStageWords;    // words of the window

    __device__ __forceinline__ void begin(uint64_t first_line, bool drop, uint32_t *bitmap) {
        word = first_line >> 5; fill = (uint32_t)first_line & 31u; drop_first = drop; out = bitmap;
    }
    __device__ __forceinline__ void begin_staged(uint64_t first_line, uint64_t window_word, bool drop, uint32_t *bitmap, uint32_t *lds) {
        word = (first_line >> 5) - window_word; fill = (uint32_t)first_line & 31u; drop_first = drop;
        out = bitmap + window_word; stage = lds;
    }
    __device__ __forceinline__ void emit() {
        if (!outw || !writer) return;
        if (STAGED && word < stage_words) atomicOr(&stage[(uint32_t)word], outw);
        else atomicOr(&out[word], outw);
    }
    __device__ __forceinline__ void push(uint32_t nl, uint32_t acc) { bits = (bits << nl) | acc; }
    __device__ __forceinline__ void flush() {
        const int n = 31 - __clz((int)bits);
        if (n > 0) {                                         // n <= 31: callers flush before bits can overflow
            uint32_t rev = __brev(bits & ((1u << n) - 1u)) >> (32 - n);      // oldest line at bit 0
            if (drop_first) { rev &= ~1u; drop_first = false; }
            outw |= rev << fill;
            uint32_t nf = fill + (uint32_t)n;
            if (nf >= 32u) {                                 // then fill >= 1
                emit();
                word++;
                outw = rev >> (32u - fill);
                nf -= 32u;
            }
            fill = nf;
            seen += (uint32_t)n;
            bits = 1;
        }
    }
    __device__ __forceinline__ void finish() {
        flush();
        emit();
        outw = 0;
    }
};
typedef ResultsT<false> Results;

// One-pass mode (rrx_match_device: no line index exists yet): a lane does not know the index of its first line, so it
// packs its verdicts from bit 0 of its OWN stream and stores the stream word by word into the workgroup's slab in HBM,
// slab[word][lane] (word k of the lanes of a workgroup is one contiguous row).  The per-stripe newline counts the lanes
// write on the side are scanned afterwards, and compact_streams_kernel shifts every lane's stream to its place in the
// accept bitmap.  bit k of a lane's stream = the k-th line end it saw (the first one belongs to the lane before if the
// stripe starts inside a line: the compaction drops it, as ResultsT does with drop_first).
struct LocalResults {
    uint32_t bits = 1, outw = 0, fill = 0, seen = 0, k = 0;
    uint32_t *__restrict__ dst;            // &slab[0][lane]
    __device__ __forceinline__ void begin(uint32_t *slab_lane) { dst = slab_lane; }
    __device__ __forceinline__ void push(uint32_t nl, uint32_t acc) { bits = (bits << nl) | acc; }
    __device__ __forceinline__ void flush() {
        const int n = 31 - __clz((int)bits);
        if (n > 0) {
            const uint32_t rev = __brev(bits & ((1u << n) - 1u)) >> (32 - n);      // oldest line at bit 0
            outw |= rev << fill;
            uint32_t nf = fill + (uint32_t)n;
            if (nf >= 32u) {
                dst[(size_t)k * kThreads] = outw;
                k++;
                outw = fill ? rev >> (32u - fill) : 0u;
                nf -= 32u;
            }
            fill = nf;
            seen += (uint32_t)n;
            bits = 1;
        }
    }
    __device__ __forceinline__ void finish() {
        flush();
        if (fill) dst[(size_t)k * kThreads] = outw;
    }
};
constexpr uint32_t kCountMask = 0x3fffffffu;             // counts[g]: bits 0..29 = '\n' in stripe g
constexpr uint32_t kExtraResult = 0x40000000u;           // bit 30 (one-pass mode): the lane's stream holds one result more than
                                                         //   its stripe has '\n' (a line followed past the stripe end, or the
                                                         //   last line of a corpus that does not end in '\n')
__host__ __device__ inline size_t slab_words_per_lane(uint32_t stripe) { return stripe / 32 + 1; }

template <class Engine, class Program>
__device__ __forceinline__ void match_stripes_body(const Program &prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                   uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                   uint32_t *__restrict__ accept_bits, uint32_t stage_off, uint32_t stage_words) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // result window of the workgroup (ResultsT<true>), behind the tables: their entries hold 16-bit LDS addresses
    uint32_t *stage = reinterpret_cast<uint32_t *>(smem + stage_off);
    Engine eng;
    eng.load(prog, smem);
    if (Engine::kStaged)
        for (uint32_t i = threadIdx.x; i < stage_words; i += kThreads) stage[i] = 0;
    __syncthreads();

    const size_t g0 = (size_t)blockIdx.x * kThreads;
    const uint64_t window_word = line_of(stripe_base[g0]) >> 5;      // the workgroup's first stripe exists: uniform load
    const size_t g = g0 + threadIdx.x;
    const size_t start = g * (size_t)stripe;
    if (start < nbytes) {                                            // (no early return: the write-out below is collective)
    const size_t stripe_end = start + stripe;
    const size_t my_end = stripe_end < nbytes ? stripe_end : nbytes;
    const uint64_t my_base = stripe_base[g];
    const bool fresh = (my_base & kFreshStripe) != 0;
    typename Engine::State st = fresh ? eng.fresh() : eng.skipping();
    ResultsT<Engine::kStaged> res;
    if (Engine::kStaged) { res.begin_staged(line_of(my_base), window_word, !fresh, accept_bits, stage); res.stage_words = stage_words; }
    else res.begin(line_of(my_base), !fresh, accept_bits);

    // ---- main phase: whole 128-byte rounds of my stripe.  The 8 loads of a line are issued as ONE burst after the
    // previous line has been consumed (they merge on one L2 request; other waves of the SIMD cover the fetch).
    // Measured alternatives, all slower: refilling each 16-byte slot right after use (one L2 request per slot),
    // a register double buffer (94 VGPRs), two half-line bursts, a software-prefetch touch.
    size_t pos = start;
    const uint4 *src = reinterpret_cast<const uint4 *>(bytes + start);
    // Engines with many words per set take half a line per round: their registers go to the state set, and they are
    // bound by the VALU, not by the second fetch of a line's other half.
    constexpr int kSlots = Engine::kRoundBytes / 16;
    const int rounds = (int)((my_end - start) / Engine::kRoundBytes);
    TextRound<kSlots> buf;                                       // named slots: never indexed at run time
    if (rounds > 0) buf.load(src);
    for (int r = 0; r < rounds; r++) {
        buf.for_each_slot([&](const uint4 &v) {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; q++) eng.consume_word(st, w[q], res.bits);
            if (res.bits >> 15) res.flush();                 // <= 16 more results fit before the next check
        });
        // All lanes flush together every 512 bytes (~11 lines of typical text fit the 31 result slots).
        // Every round costs 6 %; leaving it to the overflow check above makes the lanes flush at different
        // times, so that almost every check diverges: measured slower than either.
        if ((r & (512 / Engine::kRoundBytes - 1)) == 512 / Engine::kRoundBytes - 1) res.flush();
        if (r + 1 < rounds) buf.load(src + (size_t)(r + 1) * kSlots);
    }
    pos += (size_t)rounds * Engine::kRoundBytes;

    // ---- tail of the corpus inside my stripe (only the last stripe has one), byte by byte
    for (; pos < my_end; pos++) {
        uint32_t nl, acc;
        eng.step(st, bytes[pos], nl, acc);
        res.push(nl, acc);
        if (res.bits >> 30) res.flush();
    }
    res.flush();

    // ---- follow my last line past the stripe end.  It is mine iff I started it: I began at a line start or
    // saw a '\n' inside my stripe, and my stripe does not end exactly on a '\n'.
    const bool started = fresh || res.seen > 0;
    if (started && bytes[my_end - 1] != '\n') {
        uint32_t nl = 0, acc = 0;
        // 16 bytes per load (pos is 16-byte aligned here unless the corpus ended inside my stripe)
        while (pos + 16 <= nbytes && !nl) {
            const uint4 v = *reinterpret_cast<const uint4 *>(bytes + pos);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (!nl) eng.step(st, (w[k >> 2] >> (8 * (k & 3))) & 0xffu, nl, acc);
            pos += 16;
        }
        for (; pos < nbytes && !nl; pos++) eng.step(st, bytes[pos], nl, acc);
        if (!nl) eng.step(st, '\n', nl, acc);       // the corpus ends without '\n': end of data ends the line
        res.push(nl, acc);
    }
    res.finish();
    }
    if (Engine::kStaged) {
        __syncthreads();                             // write the window out: consecutive lanes, consecutive words
        for (uint32_t i = threadIdx.x; i < stage_words; i += kThreads) {
            const uint32_t v = stage[i];
            if (v) atomicOr(&accept_bits[window_word + i], v);
        }
    }
}

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
    __device__ void load(const Dfa2Device &p, uint16_t *p_lds, uint8_t *t_lds) {
        uint32_t *pl = reinterpret_cast<uint32_t *>(p_lds);
        const uint32_t *ps = reinterpret_cast<const uint32_t *>(p.P);
        for (int i = threadIdx.x; i < (int)(kDfa2PBytes / 4); i += blockDim.x) pl[i] = ps[i];
        uint32_t *t = reinterpret_cast<uint32_t *>(t_lds);
        const uint32_t tbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)t_lds;
        const int n = (int)(p.nrows * p.stride);
        for (int i = threadIdx.x; i < n; i += blockDim.x) t[i] = p.T2[i] + tbase;
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
};

// Same stripe geometry, feed and result path as match_stripes_kernel; pairs are aligned to even byte positions
// (stripes are even-sized), a line end may fall on either byte of a pair.
// ONEPASS (rrx_match_device): no line index yet.  Every lane starts in the start state (a lane that begins inside a line
// produces a verdict for the fragment, which the compaction drops), counts its '\n' on the side, and keeps its verdict
// stream in the workgroup's slab (LocalResults).  Bytes >= 0x80 cannot index the pair table: a text word that holds one is
// rewritten with 0x00 in their place (which rejects the line just the same) under a wave-uniform branch.
template <bool ONEPASS>
__device__ __forceinline__ void dfa2_body(const Dfa2Device &prog, const uint8_t *__restrict__ bytes, size_t nbytes, uint32_t stripe,
                                          const uint64_t *__restrict__ stripe_base, uint32_t *__restrict__ accept_bits,
                                          uint32_t *__restrict__ counts, uint32_t *__restrict__ slabs) {
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

    const size_t g0 = (size_t)blockIdx.x * kThreads;
    uint64_t window_word = 0;
    if (!ONEPASS) window_word = line_of(stripe_base[g0]) >> 5;       // the workgroup's first stripe exists: uniform load
    const size_t g = g0 + threadIdx.x;
    const size_t start = g * (size_t)stripe;
    if (start < nbytes) {                                            // (no early return: the write-out below is collective)
    const size_t stripe_end = start + stripe;
    const size_t my_end = stripe_end < nbytes ? stripe_end : nbytes;
    bool fresh = true;
    typename std::conditional<ONEPASS, LocalResults, ResultsT<true>>::type res;
    if constexpr (ONEPASS) {
        res.begin(slabs + (size_t)blockIdx.x * slab_words_per_lane(stripe) * kThreads + threadIdx.x);
    } else {
        const uint64_t my_base = stripe_base[g];
        fresh = (my_base & kFreshStripe) != 0;
        res.begin_staged(line_of(my_base), window_word, !fresh, accept_bits, stage);
        res.stage_words = stage_words;
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
    if (rounds > 0) {
#pragma unroll
        for (int i = 0; i < kSlots; i++) buf[i] = load_text(src + i);
    }
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int i = 0; i < kSlots; i++) {
            eng.consume_dword(st, clean(buf[i].x), res.bits);
            eng.consume_dword(st, clean(buf[i].y), res.bits);
            eng.consume_dword(st, clean(buf[i].z), res.bits);
            eng.consume_dword(st, clean(buf[i].w), res.bits);
            if (res.bits >> 15) res.flush();                 // <= 16 more results fit before the next check
        }
        if ((r & 3) == 3) res.flush();
        if (r + 1 < rounds) {
#pragma unroll
            for (int i = 0; i < kSlots; i++) buf[i] = load_text(src + (r + 1) * kSlots + i);
        }
    }
    pos += (size_t)rounds * kRound;
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
        if (res.bits >> 29) res.flush();
    }
    if (pos < my_end) {
        const uint32_t b = byte_at(pos);
        uint32_t lines, verdicts;
        eng.step2(st, b, '\n', lines, verdicts);
        if (b == '\n') res.push(1, verdicts >> 1);
        else { res.push(1, verdicts); closed_by_end_of_data = true; }
        pos++;
    }
    res.flush();
    const uint32_t newlines = res.seen - (closed_by_end_of_data ? 1u : 0u);      // real '\n' inside my stripe

    // ---- follow my last line past the stripe end (same ownership rule as the byte kernel), pair by pair
    if (ONEPASS && res.seen == 0) fresh = g == 0 || bytes[start - 1] == '\n';   // a stripe without any '\n': whose line is it?
    const bool started = fresh || res.seen > 0;
    bool followed = false;
    if (!closed_by_end_of_data && started && bytes[my_end - 1] != '\n') {
        uint32_t lines = 0, verdicts = 0;
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
        res.push(1, lines == 2 ? verdicts >> 1 : verdicts);     // only the first line end of the pair is mine
        followed = true;
    }
    res.finish();
    if (ONEPASS)
        counts[g] = newlines | ((followed || closed_by_end_of_data) ? kExtraResult : 0u) | (bytes[my_end - 1] == '\n' ? kEndsOnNewline : 0u);
    }
    if (!ONEPASS) {
        // ---- write the window out: consecutive lanes, consecutive words (the atomics merge into whole lines in L2;
        // the first and the last word of the window are shared with the neighbouring workgroups)
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < stage_words; i += kThreads) {
            const uint32_t v = stage[i];
            if (v) atomicOr(&accept_bits[window_word + i], v);
        }
    }
}
__global__ __launch_bounds__(kThreads) void match_stripes2_kernel(Dfa2Device prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                   uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                                   uint32_t *__restrict__ accept_bits) {
    dfa2_body<false>(prog, bytes, nbytes, stripe, stripe_base, accept_bits, nullptr, nullptr);
}
__global__ __launch_bounds__(kThreads) void match_stripes2_onepass_kernel(Dfa2Device prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                           uint32_t stripe, uint32_t *__restrict__ counts,
                                                                           uint32_t *__restrict__ slabs) {
    dfa2_body<true>(prog, bytes, nbytes, stripe, nullptr, nullptr, counts, slabs);
}

// One-pass mode, last step: lane = stripe.  The stream of stripe g (counts[g] results, the first of them dropped if the
// stripe starts inside a line) goes to bits [base, base + n) of the accept bitmap, base = '\n' before the stripe.
__global__ __launch_bounds__(256) void compact_streams_kernel(const uint32_t *__restrict__ counts, const uint64_t *__restrict__ stripe_base,
                                                               size_t nstripes, uint32_t stripe, const uint32_t *__restrict__ slabs,
                                                               uint32_t *__restrict__ accept_bits, size_t cap_words,
                                                               uint32_t *__restrict__ overflow) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= nstripes) return;
    const uint32_t c = counts[g];
    const uint32_t n = (c & kCountMask) + ((c & kExtraResult) ? 1u : 0u);
    if (!n) return;
    const uint64_t b = stripe_base[g];
    const uint64_t base = line_of(b);
    const bool fresh = (b & kFreshStripe) != 0;
    const uint32_t *src = slabs + (g / kThreads) * slab_words_per_lane(stripe) * kThreads + (g % kThreads);
    for (uint32_t k = 0; k * 32 < n; k++) {
        uint32_t v = src[(size_t)k * kThreads];
        if (n - k * 32 < 32) v &= (1u << (n - k * 32)) - 1u;
        if (k == 0 && !fresh) v &= ~1u;                          // that line belongs to the lane before me
        if (!v) continue;
        const uint64_t bit = base + (uint64_t)k * 32;
        const uint64_t word = bit >> 5;
        const uint32_t sh = (uint32_t)bit & 31u;
        if (word + (sh ? 1 : 0) >= cap_words) { atomicOr(overflow, 1u); continue; }
        atomicOr(&accept_bits[word], v << sh);
        if (sh && (v >> (32u - sh))) atomicOr(&accept_bits[word + 1], v >> (32u - sh));
    }
}

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
            uint64_t mine = __ballot((e0 | e1) != 0) & gmask;
            while (__ballot(mine != 0)) {                        // (the condition is the same in every active lane)
                const int src = mine ? __ffsll((long long)mine) - 1 : lane;
                const uint32_t w0 = __shfl(e0, src, 64), w1 = __shfl(e1, src, 64);
                if (mine) {
                    const int bit = w0 ? __ffs(w0) - 1 : 32 + __ffs(w1) - 1;
                    if (lane == src) { if (bit < 32) e0 &= ~(1u << bit); else e1 &= ~(1u << (bit - 32)); }
                    const uint2 row = X[(size_t)xidx[(src - gbase) * 64 + bit] * G + lig];
                    t0 |= row.x; t1 |= row.y;
                }
                mine = __ballot((e0 | e1) != 0) & gmask;
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

// ============================================================================================ block-cooperative NFA
// Automata beyond 4096 positions (the reference's Roaring class taken at its word, Parser.cpp:165: any size): one
// WORKGROUP holds one state set, lane l the positions [64 l, 64 l + 64) - up to 1024 lanes = 65536 positions - and
// steps one string.  What crosses waves goes through LDS, double-buffered so that a byte costs ONE barrier:
//   * the shift: every lane publishes its upper word, the lane above picks it up after the barrier;
//   * exception edges: kept SPARSE (CSR lists in HBM/L2; dense rows would take N*N/8 bytes), a lane walks the live
//     exception positions of its 64 and ORs the target bits into an LDS accumulator, which the owning lanes merge and
//     clear after the barrier - the set stays dense where it is populated, the rules that reach across it are lists;
//   * the verdict is a barrier-with-OR, only on '\n' / end of string.
// B rows are per byte class in HBM/L2, [class][lane] 8-byte words, read coalesced.  Everything is workgroup-uniform.
struct BlockNfa {
    uint32_t fin0, fin1, self0, self1, exc0, exc1;
    const uint2 *__restrict__ rows;        // HBM/L2 [ncls][T]
    const uint8_t *cls;                    // LDS [256]
    const uint32_t *__restrict__ xoff, *__restrict__ xtgt;
    uint32_t *top, *texc;                  // LDS: [2][T], [2][2T]
    bool any_exc;
    uint32_t par = 0;
    int lane, T;

    static size_t lds_bytes(uint32_t T) { return (size_t)T * 4 * 6 + 256; }
    __device__ void load(const BlockNfaDevice &p, uint8_t *lds, bool line_mode) {
        lane = threadIdx.x; T = blockDim.x;
        top = reinterpret_cast<uint32_t *>(lds);
        texc = top + 2 * T;
        uint8_t *c = reinterpret_cast<uint8_t *>(texc + 4 * T);
        const uint8_t *src = line_mode ? p.cls_line : p.cls_plain;
        for (int i = lane; i < 256; i += T) c[i] = src[i];
        for (int i = lane; i < 4 * T; i += T) texc[i] = 0;
        cls = c;
        const uint2 *m = reinterpret_cast<const uint2 *>(p.masks);
        uint2 v;
        v = m[0 * T + lane]; fin0 = v.x; fin1 = v.y;
        v = m[1 * T + lane]; self0 = v.x; self1 = v.y;
        v = m[2 * T + lane]; exc0 = v.x; exc1 = v.y;
        rows = reinterpret_cast<const uint2 *>(p.Bcls); xoff = p.xoff; xtgt = p.xtgt; any_exc = p.any_exc != 0;
    }
    // called by the whole workgroup (it is a barrier)
    __device__ __forceinline__ bool accepting(uint32_t s0, uint32_t s1) const {
        return __syncthreads_or((((s0 & fin0) | (s1 & fin1)) != 0) ? 1 : 0) != 0;
    }
    __device__ __forceinline__ void scatter(uint32_t e, uint32_t base, uint32_t *acc) const {
        while (e) {
            const uint32_t p = base + (uint32_t)__ffs(e) - 1u;
            e &= e - 1;
            for (uint32_t k = xoff[p], hi = xoff[p + 1]; k < hi; k++) { const uint32_t q = xtgt[k]; atomicOr(&acc[q >> 5], 1u << (q & 31)); }
        }
    }
    // c is the same in every lane of the workgroup
    template <bool LINE>
    __device__ __forceinline__ void advance(uint32_t &s0, uint32_t &s1, uint32_t c) {
        const uint2 b = rows[(size_t)cls[c] * T + lane];
        uint32_t *tp = top + par * T, *acc = texc + par * 2 * T;
        tp[lane] = s1;
        if (any_exc) { scatter(s0 & exc0, (uint32_t)lane * 64u, acc); scatter(s1 & exc1, (uint32_t)lane * 64u + 32u, acc); }
        __syncthreads();
        const uint32_t below = lane ? tp[lane - 1] : (LINE ? 0x80000000u : 0u);
        uint32_t t0 = __builtin_amdgcn_alignbit(s0, below, 31) | (s0 & self0);
        uint32_t t1 = __builtin_amdgcn_alignbit(s1, s0, 31) | (s1 & self1);
        if (any_exc) { t0 |= acc[2 * lane]; t1 |= acc[2 * lane + 1]; acc[2 * lane] = 0; acc[2 * lane + 1] = 0; }
        s0 = t0 & b.x; s1 = t1 & b.y;
        par ^= 1u;
    }
};

// One workgroup per stripe; every lane mirrors the (uniform) result bookkeeping, lane 0 writes.
__global__ __launch_bounds__(1024) void match_stripes_block_kernel(BlockNfaDevice prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                    uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                                    uint32_t *__restrict__ accept_bits) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    BlockNfa eng;
    eng.load(prog, smem, true);
    __syncthreads();
    const size_t g = blockIdx.x;
    const size_t start = g * (size_t)stripe;
    if (start >= nbytes) return;
    const size_t stripe_end = start + stripe;
    const size_t my_end = stripe_end < nbytes ? stripe_end : nbytes;
    const uint64_t my_base = stripe_base[g];
    const bool fresh = (my_base & kFreshStripe) != 0;
    uint32_t s0 = (fresh && eng.lane == 0) ? 1u : 0u, s1 = 0;
    Results res;
    res.begin(line_of(my_base), !fresh, accept_bits);
    res.writer = eng.lane == 0;
    size_t pos = start;
    for (; pos < my_end; pos++) {
        const uint32_t c = bytes[pos];
        if (c == '\n') { res.push(1, eng.accepting(s0, s1) ? 1u : 0u); if (res.bits >> 30) res.flush(); }
        eng.advance<true>(s0, s1, c);
    }
    res.flush();
    const bool started = fresh || res.seen > 0;
    if (started && bytes[my_end - 1] != '\n') {
        bool ended = false;
        for (; pos < nbytes && !ended; pos++) {
            const uint32_t c = bytes[pos];
            if (c == '\n') ended = true;
            else eng.advance<true>(s0, s1, c);
        }
        res.push(1, eng.accepting(s0, s1) ? 1u : 0u);
    }
    res.finish();
}
__global__ __launch_bounds__(1024) void match_extents_block_kernel(BlockNfaDevice prog, const uint8_t *__restrict__ bytes,
                                                                    const uint64_t *__restrict__ off, size_t nitems, uint32_t trim,
                                                                    uint8_t *__restrict__ accept) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    BlockNfa eng;
    eng.load(prog, smem, false);
    __syncthreads();
    const size_t i = blockIdx.x;
    if (i >= nitems) return;
    size_t b = off[i], e = off[i + 1];
    e = e - b >= trim ? e - trim : b;
    uint32_t s0 = eng.lane == 0 ? 1u : 0u, s1 = 0;
    for (size_t pos = b; pos < e; pos++) eng.advance<false>(s0, s1, bytes[pos]);
    const bool ok = eng.accepting(s0, s1);
    if (eng.lane == 0) accept[i] = ok ? 1 : 0;
}

// ============================================================================================ extents kernel
// One lane per item; bytes come straight from HBM/L2.  Used for explicit (offset,len) batches, for the
// iterator facade's single strings, and wherever '\n' is an ordinary character.
// Two entry points for one body.  The table engines run best as the compiler allocates them (66 VGPRs; capping them at 64
// for a second workgroup per CU measured -6 %); the register-resident NFA engines gain from the cap (+2 ... +13 %,
// W >= 2 spills a little to scratch).
template <class Engine, class Program>
__global__ __launch_bounds__(kThreads) void match_stripes_kernel(Program prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                  uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                                  uint32_t *__restrict__ accept_bits, uint32_t stage_off,
                                                                  uint32_t stage_words) {
    match_stripes_body<Engine, Program>(prog, bytes, nbytes, stripe, stripe_base, accept_bits, stage_off, stage_words);
}
template <class Engine, class Program>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(8, 8)))
void match_stripes_kernel_8waves(Program prog, const uint8_t *__restrict__ bytes, size_t nbytes, uint32_t stripe,
                                 const uint64_t *__restrict__ stripe_base, uint32_t *__restrict__ accept_bits, uint32_t stage_off,
                                 uint32_t stage_words) {
    match_stripes_body<Engine, Program>(prog, bytes, nbytes, stripe, stripe_base, accept_bits, stage_off, stage_words);
}
template <class Engine, class Program>
__global__ __launch_bounds__(kThreads) void match_extents_kernel(Program prog, const uint8_t *__restrict__ bytes,
                                                                  const uint64_t *__restrict__ off, size_t nitems, uint32_t trim,
                                                                  uint8_t *__restrict__ accept) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Engine eng;
    eng.load(prog, smem);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= nitems) return;
    size_t b = off[i], e = off[i + 1];
    e = e - b >= trim ? e - trim : b;
    typename Engine::State st;
    eng.reset(st);
    bool dead = false;
    size_t p = b;
    auto one = [&](uint32_t c) {
        if (c == 0 || c >= 0x80) { eng.kill(st); dead = true; }
        else eng.step(st, c);
    };
    for (; p < e && (p & 15) && !dead; p++) one(bytes[p]);                 // up to 16-byte alignment
    for (; p + 16 <= e && !dead; p += 16) {                                // 16 bytes per load
        const uint4 v = *reinterpret_cast<const uint4 *>(bytes + p);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (!dead) one((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
    }
    for (; p < e && !dead; p++) one(bytes[p]);
    accept[i] = eng.accepting(st) ? 1 : 0;
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
__global__ __launch_bounds__(1024) void scan_sums_kernel(uint64_t *__restrict__ sums, size_t nchunks, uint64_t *__restrict__ total) {
    __shared__ uint64_t sh[1024];
    const size_t per = (nchunks + 1023) / 1024;
    const size_t lo = threadIdx.x * per < nchunks ? threadIdx.x * per : nchunks, hi = lo + per < nchunks ? lo + per : nchunks;
    uint64_t s = 0;
    for (size_t i = lo; i < hi; i++) s += sums[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; i++) { uint64_t v = sh[i]; sh[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    uint64_t run = sh[threadIdx.x];
    for (size_t i = lo; i < hi; i++) { uint64_t v = sums[i]; sums[i] = run; run += v; }
}
__global__ __launch_bounds__(256) void scan_chunks_kernel(const uint32_t *__restrict__ counts, size_t n, const uint64_t *__restrict__ sums,
                                                           uint64_t *__restrict__ base) {
    __shared__ uint64_t sh[256];
    const size_t lo = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * (kScanChunk / 256);
    uint64_t s = 0;
    for (size_t i = lo; i < lo + kScanChunk / 256 && i < n; i++) s += counts[i] & kCountMask;
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = sums[blockIdx.x];
        for (int i = 0; i < 256; i++) { uint64_t v = sh[i]; sh[i] = run; run += v; }
    }
    __syncthreads();
    uint64_t run = sh[threadIdx.x];
    for (size_t i = lo; i < lo + kScanChunk / 256 && i < n; i++) {
        const bool fresh = i == 0 || (counts[i - 1] & kEndsOnNewline);
        base[i] = run | (fresh ? kFreshStripe : 0);
        run += counts[i] & kCountMask;
    }
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel, device and size increase, not once per launch.
// (One slot array per kernel: `slots` is a function-local static of the calling template instantiation.)
constexpr int kMaxDevices = 64;
struct LdsAttr { std::atomic<int> bytes[kMaxDevices]; };
inline hipError_t ensure_dynamic_lds(LdsAttr &slots, const void *kernel, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (slots.bytes[dev].load(std::memory_order_acquire) >= (int)bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) slots.bytes[dev].store((int)bytes, std::memory_order_release);
    return e;
}

template <class Engine, class Program>
int launch_stripes(const Program &p, size_t table_bytes, const uint8_t *bytes, size_t nbytes, uint32_t stripe,
                   const uint64_t *stripe_base, size_t nstripes, uint32_t *accept, void *stream) {
    if (!nstripes) return 0;
    void (*k)(Program, const uint8_t *, size_t, uint32_t, const uint64_t *, uint32_t *, uint32_t, uint32_t);
    if constexpr (Engine::kEightWaves) k = match_stripes_kernel_8waves<Engine, Program>;
    else k = match_stripes_kernel<Engine, Program>;
    const uint32_t stage_off = (uint32_t)((table_bytes + 15) & ~(size_t)15);
    // the window takes what the tables leave of half a CU's LDS (two workgroups per CU), 16 KiB at least
    const size_t half_cu = 80 * 1024;
    const uint32_t stage_words = stage_off + kStageWords * sizeof(uint32_t) >= half_cu ? kStageWords : (uint32_t)((half_cu - stage_off) / 4);
    const size_t lds = Engine::kStaged ? stage_off + (size_t)stage_words * sizeof(uint32_t) : table_bytes;
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(k), lds);
    if (e != hipSuccess) return (int)e;
    size_t blocks = (nstripes + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), lds, (hipStream_t)stream, p, bytes, nbytes, stripe, stripe_base, accept, stage_off, stage_words);
    return (int)hipGetLastError();
}

// ============================================================================================ search
// Two kernels.  line_offsets_kernel (once per corpus): lane = stripe, every '\n' at p inside the stripe starts the next
// line at p + 1 (line numbers from the stripe index).  search_lines_kernel: lane = line.  Step the forward DFA until it
// accepts (that is the smallest match end e) and stop there - a lane never reads the rest of its line - then walk the
// reverse DFA back from e to the line start, remembering the last position where it accepts (the smallest start of a
// match that ends at e).  Consecutive lanes own consecutive lines: their text is contiguous and both result arrays are
// written coalesced.
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

struct SearchTables {
    const uint8_t *cls;
    const uint16_t *next_f, *next_r;                     // bit 15 of an entry = the state it leads to is accepting
    uint32_t K, start_f, start_r;
    bool empty_matches;
    __device__ void load(const SearchDevice &p, uint8_t *lds) {
        uint16_t *nf = reinterpret_cast<uint16_t *>(lds);
        uint16_t *nr = nf + (size_t)p.nf * p.ncls;
        uint8_t *c = reinterpret_cast<uint8_t *>(nr + (size_t)p.nr * p.ncls);
        for (uint32_t i = threadIdx.x; i < p.nf * p.ncls; i += blockDim.x) { const uint16_t t = p.next_f[i]; nf[i] = (uint16_t)(t | (p.acc_f[t] ? 0x8000u : 0u)); }
        for (uint32_t i = threadIdx.x; i < p.nr * p.ncls; i += blockDim.x) { const uint16_t t = p.next_r[i]; nr[i] = (uint16_t)(t | (p.acc_r[t] ? 0x8000u : 0u)); }
        for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) c[i] = p.cls[i];
        cls = c; next_f = nf; next_r = nr; K = p.ncls; start_f = p.start_f; start_r = p.start_r;
        empty_matches = p.acc_f[p.start_f] != 0;
    }
};
// First match of bytes[p, b) whose start is >= p: false if there is none.  s, e are absolute offsets.
__device__ __forceinline__ bool search_from(const SearchTables &t, const uint8_t *__restrict__ bytes, size_t nbytes, size_t p, size_t b,
                                            size_t &s, size_t &e) {
    if (t.empty_matches) { s = p; e = p; return true; }          // the pattern accepts "": the match is [p, p)
    uint32_t q = t.start_f;
    size_t pos = p;
    bool found = false;
    while (pos < b && !found) {                                   // aligned dwords; bytes before `pos` / from `b` on are skipped
        const size_t base = pos & ~(size_t)3;
        uint32_t w;
        if (base + 4 <= nbytes) w = *reinterpret_cast<const uint32_t *>(bytes + base);
        else { w = 0; for (size_t k = pos; k < nbytes; k++) w |= (uint32_t)bytes[k] << (8 * (k - base)); }     // last dword of the data
        const size_t stop = base + 4 < b ? base + 4 : b;
        for (; pos < stop; pos++) {
            const uint32_t x = t.next_f[q * t.K + t.cls[(w >> (8 * (pos - base))) & 0xffu]];
            q = x & 0x7fffu;
            if (x & 0x8000u) { found = true; pos++; break; }
        }
    }
    if (!found) return false;
    e = pos;
    uint32_t r = t.start_r;
    size_t best = pos;
    for (size_t k = pos; k > p;) {
        k--;
        const uint32_t x = t.next_r[r * t.K + t.cls[bytes[k]]];
        r = x & 0x7fffu;
        if (!r) break;                                            // state 0 is dead
        if (x & 0x8000u) best = k;
    }
    s = best;
    return true;
}
__global__ __launch_bounds__(256) void search_lines_kernel(SearchDevice prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                           const uint64_t *__restrict__ line_off, size_t nlines,
                                                           uint32_t *__restrict__ match_start, uint32_t *__restrict__ match_end) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    SearchTables t;
    t.load(prog, smem);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlines) return;
    const size_t a = line_off[i], b = line_off[i + 1] - 1;        // [a, b): the line without its '\n'
    size_t s, e;
    const bool found = search_from(t, bytes, nbytes, a, b, s, e);
    match_start[i] = found ? (uint32_t)(s - a) : 0xffffffffu;
    match_end[i] = found ? (uint32_t)(e - a) : 0xffffffffu;
}
// All matches of a line, left to right: after a match the search continues at its end (one byte further after an empty
// match).  FILL = false: count[i] = number of matches.  FILL = true: the matches of line i go to slots first[i], ...
template <bool FILL>
__global__ __launch_bounds__(256) void search_all_kernel(SearchDevice prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                         const uint64_t *__restrict__ line_off, size_t nlines,
                                                         uint32_t *__restrict__ count, const uint64_t *__restrict__ first,
                                                         uint32_t *__restrict__ match_start, uint32_t *__restrict__ match_end) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    SearchTables t;
    t.load(prog, smem);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlines) return;
    const size_t a = line_off[i], b = line_off[i + 1] - 1;
    uint64_t slot = FILL ? first[i] : 0;
    uint32_t k = 0;
    for (size_t p = a; p <= b;) {
        size_t s, e;
        if (!search_from(t, bytes, nbytes, p, b, s, e)) break;
        if (FILL) { match_start[slot + k] = (uint32_t)(s - a); match_end[slot + k] = (uint32_t)(e - a); }
        k++;
        p = e > s ? e : e + 1;
    }
    if (!FILL) count[i] = k;
}

// ============================================================================================ one long string
// Chunk maps.  LDS: the plain DFA widened to one u16 entry per (state, byte value 0..127 | >= 0x80), entry = row
// offset of the next state (state * 129), so a step is one clamp, one add and one ds_read_u16.
constexpr int kLongThreads = 256;
__global__ __launch_bounds__(kLongThreads) void long_maps_kernel(DfaDevice p, const uint8_t *__restrict__ bytes, size_t nbytes, uint32_t chunk,
                                                                 uint32_t nchunks, uint16_t *__restrict__ maps) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *wide = reinterpret_cast<uint16_t *>(smem);
    const uint32_t D = p.nstates;
    for (uint32_t i = threadIdx.x; i < D * kWideColumns; i += kLongThreads) {
        const uint32_t s = i / kWideColumns, c = i % kWideColumns;
        wide[i] = (uint16_t)(p.next[s * p.ncls + p.cls[c]] * kWideColumns);     // column 128 stands for every byte >= 0x80
    }
    __syncthreads();
    const uint32_t per_block = kLongThreads / D, ci = threadIdx.x / D, s0 = threadIdx.x % D;
    const size_t k = (size_t)blockIdx.x * per_block + ci;
    if (ci >= per_block || k >= nchunks) return;
    const size_t a = k * (size_t)chunk, b = a + chunk < nbytes ? a + chunk : nbytes;
    uint32_t row = s0 * kWideColumns;
    size_t pos = a;
    if ((reinterpret_cast<uintptr_t>(bytes) & 15) == 0) {                       // chunk starts are multiples of 16
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
    maps[k * D + s0] = (uint16_t)(row / kWideColumns);
}
// out[g] = in[g*group + group-1] o ... o in[g*group]   (one lane per start state; dependent L2 reads)
__global__ __launch_bounds__(kLongThreads) void long_compose_kernel(const uint16_t *__restrict__ in, uint32_t nin, uint32_t D, uint32_t group,
                                                                    uint16_t *__restrict__ out) {
    const uint32_t j = threadIdx.x;
    if (j >= D) return;
    const size_t lo = (size_t)blockIdx.x * group, hi = lo + group < nin ? lo + group : nin;
    uint32_t s = j;
    for (size_t k = lo; k < hi; k++) s = in[k * D + s];
    out[(size_t)blockIdx.x * D + j] = (uint16_t)s;
}
__global__ void long_finish_kernel(const uint16_t *__restrict__ map, DfaDevice p, uint8_t *__restrict__ accept) {
    if (threadIdx.x == 0 && blockIdx.x == 0) accept[0] = p.acc[map[p.start]];
}

template <class Engine, class Program>
int launch_extents(const Program &p, size_t table_bytes, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim,
                   uint8_t *accept, void *stream) {
    if (!nitems) return 0;
    auto k = match_extents_kernel<Engine, Program>;
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(k), table_bytes);
    if (e != hipSuccess) return (int)e;
    size_t blocks = (nitems + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), table_bytes, (hipStream_t)stream, p, bytes, off, nitems, trim, accept);
    return (int)hipGetLastError();
}

}  // namespace

int count_newlines_per_stripe(const uint8_t *bytes, size_t nbytes, uint32_t stripe, uint32_t *counts, size_t nstripes, uint32_t *flags,
                              void *stream) {
    if (!nstripes) return 0;
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

#define RRX_NFA_DISPATCH(CALL)                                                      \
    switch (p.W) {                                                                  \
    case 1: return CALL(1);                                                         \
    case 2: return CALL(2);                                                         \
    case 3: return CALL(3);                                                         \
    case 4: return CALL(4);                                                         \
    case 5: case 6: return CALL(6);                                                 \
    case 7: case 8: return CALL(8);                                                 \
    case 9: case 10: case 11: case 12: return CALL(12);                             \
    case 13: case 14: case 15: case 16: return CALL(16);                            \
    default: return (int)hipErrorInvalidValue;                                      \
    }

// The device tables are padded to the instantiated width by the caller (NfaDevice::W is the padded width).
int match_stripes_nfa(const NfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                      size_t nstripes, uint32_t *accept, void *stream) {
    // four builds per width: the bare chain, + self loops, + add-carry groups, + exception rows
#define GO(WW, S, R, C) launch_stripes<LineNfaEngine<WW, S, R, C>, NfaDevice>(p, LineNfaEngine<WW, S, R, C>::lds_bytes(p), bytes, nbytes, stripe, stripe_base, nstripes, accept, stream)
#define CALL(WW) (p.any_exc ? GO(WW, true, true, false) : p.any_carry ? GO(WW, true, false, true) : p.any_self ? GO(WW, true, false, false) : GO(WW, false, false, false))
    RRX_NFA_DISPATCH(CALL)
#undef CALL
#undef GO
}
int match_stripes_dfa(const LineDfaDevice &p, bool clamp_high, const uint8_t *bytes, size_t nbytes, uint32_t stripe,
                      const uint64_t *stripe_base, size_t nstripes, uint32_t *accept, void *stream) {
#define GO(WIDE, CLAMP) launch_stripes<LineDfaEngine<WIDE, CLAMP>, LineDfaDevice>(p, LineDfaEngine<WIDE, CLAMP>::lds_bytes(p), bytes, nbytes, stripe, stripe_base, nstripes, accept, stream)
    if (p.in_global) return launch_stripes<LineDfaGlobalEngine, LineDfaDevice>(p, 256, bytes, nbytes, stripe, stripe_base, nstripes, accept, stream);
    if (p.wide) return clamp_high ? GO(true, true) : GO(true, false);
    return GO(false, false);
#undef GO
}
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
int match_stripes_block_nfa(const BlockNfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                            size_t nstripes, uint32_t *accept, void *stream) {
    if (!nstripes) return 0;
    hipLaunchKernelGGL(match_stripes_block_kernel, dim3((unsigned)nstripes), dim3(p.T), BlockNfa::lds_bytes(p.T), (hipStream_t)stream, p, bytes, nbytes,
                       stripe, stripe_base, accept);
    return (int)hipGetLastError();
}
int match_extents_block_nfa(const BlockNfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                            void *stream) {
    if (!nitems) return 0;
    hipLaunchKernelGGL(match_extents_block_kernel, dim3((unsigned)nitems), dim3(p.T), BlockNfa::lds_bytes(p.T), (hipStream_t)stream, p, bytes, off, nitems,
                       trim, accept);
    return (int)hipGetLastError();
}
int match_stripes_dfa2(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                       size_t nstripes, uint32_t *accept, void *stream) {
    if (!nstripes) return 0;
    if (Dfa2::lds_bytes(p) > kDfa2MaxTable) return (int)hipErrorInvalidValue;
    size_t blocks = (nstripes + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(match_stripes2_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, p, bytes, nbytes, stripe, stripe_base, accept);
    return (int)hipGetLastError();
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
                    uint32_t *accept_bits, size_t cap_words, uint32_t *overflow, void *stream) {
    if (!nstripes) return 0;
    hipLaunchKernelGGL(compact_streams_kernel, dim3((unsigned)((nstripes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, counts, stripe_base, nstripes,
                       stripe, slabs, accept_bits, cap_words, overflow);
    return (int)hipGetLastError();
}
int match_extents_nfa(const NfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                      void *stream) {
#define CALL(WW) launch_extents<PlainNfaEngine<WW>, NfaDevice>(p, PlainNfaEngine<WW>::lds_bytes(p), bytes, off, nitems, trim, accept, stream)
    RRX_NFA_DISPATCH(CALL)
#undef CALL
}
size_t search_lds_bytes(const SearchDevice &p) { return ((size_t)p.nf + p.nr) * p.ncls * sizeof(uint16_t) + 256; }
int build_line_offsets(const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base, size_t nstripes,
                       uint64_t *line_off, void *stream) {
    if (!nstripes) return 0;
    hipLaunchKernelGGL(line_offsets_kernel, dim3((unsigned)((nstripes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bytes, nbytes, stripe,
                       stripe_base, nstripes, line_off);
    return (int)hipGetLastError();
}
int search_lines(const SearchDevice &p, const uint8_t *bytes, size_t nbytes, const uint64_t *line_off, size_t nlines, uint32_t *match_start,
                 uint32_t *match_end, void *stream) {
    if (!nlines) return 0;
    const size_t lds = search_lds_bytes(p);
    if (lds > kSearchLdsBudget || p.nf > 32767 || p.nr > 32767) return (int)hipErrorInvalidValue;
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(search_lines_kernel), lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(search_lines_kernel, dim3((unsigned)((nlines + 255) / 256)), dim3(256), lds, (hipStream_t)stream, p, bytes, nbytes, line_off, nlines,
                       match_start, match_end);
    return (int)hipGetLastError();
}
int search_all(const SearchDevice &p, const uint8_t *bytes, size_t nbytes, const uint64_t *line_off, size_t nlines, uint32_t *count,
               const uint64_t *first, uint32_t *match_start, uint32_t *match_end, void *stream) {
    if (!nlines) return 0;
    const size_t lds = search_lds_bytes(p);
    if (lds > kSearchLdsBudget || p.nf > 32767 || p.nr > 32767) return (int)hipErrorInvalidValue;
    const bool fill = first != nullptr;
    static LdsAttr attr_fill, attr_count;
    hipError_t e = fill ? ensure_dynamic_lds(attr_fill, reinterpret_cast<const void *>(search_all_kernel<true>), lds)
                        : ensure_dynamic_lds(attr_count, reinterpret_cast<const void *>(search_all_kernel<false>), lds);
    if (e != hipSuccess) return (int)e;
    const dim3 grid((unsigned)((nlines + 255) / 256));
    if (fill) hipLaunchKernelGGL(search_all_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, p, bytes, nbytes, line_off, nlines, count, first, match_start, match_end);
    else hipLaunchKernelGGL(search_all_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, p, bytes, nbytes, line_off, nlines, count, first, match_start, match_end);
    return (int)hipGetLastError();
}
static uint32_t long_chunk(size_t nbytes) {
    uint32_t chunk = 1024;                                     // at most 65536 chunks, of 1 KiB or more
    while (((nbytes + chunk - 1) / chunk) > 65536) chunk <<= 1;
    return chunk;
}
size_t long_scratch_bytes(uint32_t nstates, size_t nbytes, uint32_t *chunk) {
    *chunk = long_chunk(nbytes);
    const size_t k0 = (nbytes + *chunk - 1) / *chunk, k1 = (k0 + kLongGroup - 1) / kLongGroup;
    return (k0 + k1 + 2) * nstates * sizeof(uint16_t);         // level 0, then the levels ping-pong between two areas
}
int match_long_dfa(const DfaDevice &p, const uint8_t *bytes, size_t nbytes, uint32_t chunk, void *scratch, uint8_t *accept,
                   void *stream) {
    const uint32_t D = p.nstates;
    if (!D || D > kLongMaxStates || !nbytes) return (int)hipErrorInvalidValue;
    const size_t lds = (size_t)D * kWideColumns * sizeof(uint16_t);
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(long_maps_kernel), lds);
    if (e != hipSuccess) return (int)e;
    uint32_t n = (uint32_t)((nbytes + chunk - 1) / chunk);
    uint16_t *cur = static_cast<uint16_t *>(scratch), *other = cur + (size_t)n * D;
    const uint32_t per_block = kLongThreads / D;
    hipLaunchKernelGGL(long_maps_kernel, dim3((n + per_block - 1) / per_block), dim3(kLongThreads), lds, (hipStream_t)stream, p, bytes, nbytes,
                       chunk, n, cur);
    uint16_t *area[2] = {other, cur};                          // level 1 writes behind level 0, level 2 over level 0, ...
    for (int lvl = 0; n > 1; lvl++) {
        const uint32_t m = (n + kLongGroup - 1) / kLongGroup;
        uint16_t *dst = area[lvl & 1];
        hipLaunchKernelGGL(long_compose_kernel, dim3(m), dim3(kLongThreads), 0, (hipStream_t)stream, cur, n, D, kLongGroup, dst);
        cur = dst;
        n = m;
    }
    hipLaunchKernelGGL(long_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cur, p, accept);
    return (int)hipGetLastError();
}
int match_extents_dfa(const DfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                      void *stream) {
    // tables beyond the LDS budget (the batch kernel's "global" form) stay in HBM/L2 here too
    if (PlainDfaEngine::lds_bytes(p) > kPlainDfaLdsBudget)
        return launch_extents<PlainDfaGlobalEngine, DfaDevice>(p, PlainDfaGlobalEngine::lds_bytes(p), bytes, off, nitems, trim, accept, stream);
    return launch_extents<PlainDfaEngine, DfaDevice>(p, PlainDfaEngine::lds_bytes(p), bytes, off, nitems, trim, accept, stream);
}

}  // namespace dev
}  // namespace rrx
