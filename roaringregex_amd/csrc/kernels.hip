// kernels.hip — hand-written CDNA4 (gfx950) kernels for the RoaringRegex hot path.
//
// Replaces, for a whole batch of '\n'-delimited strings at once:
//   AcceptanceIterator::operator++(int)   regex.h:156-159   (consume the string)
//   Processor::shift<true>                NFA.cc:72-102     (per-byte state-set transition)
//   Processor::operator*()                NFA.cc:103-107    (accepting?)
// Pure integer/bitwise work, HBM-read bound by design: no MFMA.
//
// Batch kernel (match_stripes<Engine>) — the text never touches LDS:
//   * lane g of the grid owns the lines that START in its contiguous 4 KiB stripe of the corpus and follows
//     its last line past the stripe end, so every line is stepped by exactly one lane from its first byte;
//   * each lane streams its own stripe from HBM straight into registers, 64 bytes (4 x global_load_dwordx4)
//     per round, the next round's loads in flight while the current one is stepped (measured: this per-lane
//     streaming pattern reads at the same 6.3-6.4 TB/s as a fully coalesced copy, profiles/r01_membench*);
//   * the automaton tables live in LDS, the state lives in registers;
//   * line results are accumulated in a register as ordered bits and flushed once per round as byte stores
//     accept[line]; line = stripe_base[g] + newlines seen so far, stripe_base being the per-stripe newline
//     prefix the corpus carries (8 bytes per 4 KiB of text): no per-line offset array is ever read.
#include <hip/hip_runtime.h>

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
template <bool WIDE>
struct LineDfaEngine {
    struct State { uint32_t e; };          // last table entry; low 30 bits = current row byte offset
    const uint8_t *tab;                    // LDS, byte-addressed
    const uint8_t *cls;                    // LDS [256] (classed form)
    uint32_t start_off;

    static size_t lds_bytes(const LineDfaDevice &p) { return (size_t)p.nrows * p.stride * 4 + (WIDE ? 0 : 256); }
    __device__ void load(const LineDfaDevice &p, uint8_t *lds) {
        uint32_t *t = reinterpret_cast<uint32_t *>(lds);
        const int n = (int)(p.nrows * p.stride);
        for (int i = threadIdx.x; i < n; i += blockDim.x) t[i] = p.table[i];
        if (!WIDE) {
            uint8_t *c = lds + (size_t)n * 4;
            for (int i = threadIdx.x; i < 256; i += blockDim.x) c[i] = p.cls[i];
            cls = c;
        }
        tab = lds;
        start_off = p.start_off;
    }
    __device__ __forceinline__ State fresh() const { return State{start_off}; }
    __device__ __forceinline__ State skipping() const { return State{0}; }          // dead row: waits for '\n'
    __device__ __forceinline__ void step(State &st, uint32_t c, uint32_t &nl, uint32_t &acc) const {
        uint32_t col;
        if (WIDE) col = c < 128u ? c : 128u;
        else col = cls[c];
        const uint32_t off = (st.e & 0x3fffffffu) + col * 4u;
        st.e = *reinterpret_cast<const uint32_t *>(tab + off);
        nl = (st.e >> 30) & 1u;
        acc = st.e >> 31;
    }
};

// ---- shift-and NFA: state set in W registers --------------------------------------------------------
template <int W>
struct NfaCore {
    struct State { uint32_t s[W]; };
    const uint32_t *B;      // LDS [256][W]
    const uint32_t *X;      // LDS [nbits][W]
    NfaMasks m;
    bool any_exc;

    static size_t lds_bytes(const NfaDevice &p) { return ((size_t)256 * W + (size_t)p.nbits * W) * 4; }
    __device__ void load(const NfaDevice &p, uint8_t *lds) {
        uint32_t *b = reinterpret_cast<uint32_t *>(lds);
        uint32_t *x = b + 256 * W;
        for (int i = threadIdx.x; i < 256 * W; i += blockDim.x) b[i] = p.B[i];
        for (int i = threadIdx.x; i < (int)p.nbits * W; i += blockDim.x) x[i] = p.X[i];
        B = b; X = x; m = p.masks; any_exc = p.any_exc != 0;
    }
    __device__ __forceinline__ bool accepting(const State &st) const {
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < W; i++) a |= st.s[i] & m.fin[i];
        return a != 0;
    }
    // next = ( ((S << 1) & CHAIN) | (S & SELF) | OR_{e in S & EXC} X[e] ) & B[c]
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

template <int W>
struct LineNfaEngine : NfaCore<W> {
    using State = typename NfaCore<W>::State;
    __device__ __forceinline__ State fresh() const {
        State st;
#pragma unroll
        for (int i = 0; i < W; i++) st.s[i] = this->m.init[i];
        return st;
    }
    __device__ __forceinline__ State skipping() const {
        State st;
#pragma unroll
        for (int i = 0; i < W; i++) st.s[i] = 0;
        return st;
    }
    __device__ __forceinline__ void step(State &st, uint32_t c, uint32_t &nl, uint32_t &acc) const {
        const bool isnl = c == '\n';
        const bool a = this->accepting(st);
        this->advance(st, c);                      // B['\n'] and B[0], B[>=0x80] are empty rows: the set dies
        if (isnl) {
#pragma unroll
            for (int i = 0; i < W; i++) st.s[i] = this->m.init[i];
        }
        nl = isnl ? 1u : 0u;
        acc = (isnl && a) ? 1u : 0u;
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

// ============================================================================================ batch kernel
// Ordered line results of one lane: `bits` holds a sentinel 1 followed by one verdict bit per finished line
// (oldest highest).  flush() stores them as bytes accept[line++]; the first result of a lane that started
// inside somebody else's line belongs to that somebody and is dropped (it is still counted).
struct Results {
    uint32_t bits = 1;
    uint64_t line;
    bool drop_first;
    uint8_t *__restrict__ accept;

    __device__ __forceinline__ void push(uint32_t nl, uint32_t acc) { bits = (bits << nl) | acc; }
    __device__ __forceinline__ void flush() {
        int n = 31 - __clz((int)bits);
        while (__any(n > 0)) {
            if (n > 0) {
                n--;
                if (drop_first) drop_first = false;
                else accept[line] = (uint8_t)((bits >> n) & 1u);
                line++;
            }
        }
        bits = 1;
    }
    // same, for code that runs with lanes diverged (no wave-wide vote)
    __device__ __forceinline__ void flush_lane() {
        for (int n = 31 - __clz((int)bits); n > 0;) {
            n--;
            if (drop_first) drop_first = false;
            else accept[line] = (uint8_t)((bits >> n) & 1u);
            line++;
        }
        bits = 1;
    }
};

template <class Engine, class Program>
__global__ __launch_bounds__(kThreads) void match_stripes_kernel(Program prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                  const uint64_t *__restrict__ stripe_base,
                                                                  uint8_t *__restrict__ accept) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Engine eng;
    eng.load(prog, smem);
    __syncthreads();

    const size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x;
    const size_t start = g * (size_t)kStripe;
    if (start >= nbytes) return;
    const size_t stripe_end = start + kStripe;
    const size_t my_end = stripe_end < nbytes ? stripe_end : nbytes;
    const bool fresh = start == 0 || bytes[start - 1] == '\n';
    typename Engine::State st = fresh ? eng.fresh() : eng.skipping();
    Results res;
    res.line = stripe_base[g];
    res.drop_first = !fresh;
    res.accept = accept;
    const uint64_t first_line = res.line;

    // ---- main phase: whole 64-byte rounds of my stripe, next round's loads in flight
    size_t pos = start;
    const uint4 *src = reinterpret_cast<const uint4 *>(bytes + start);
    const int rounds = (int)((my_end - start) / kRound);
    uint4 cur[4], nxt[4];
    if (rounds > 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) cur[i] = src[i];
    }
    for (int r = 0; r < rounds; r++) {
        if (r + 1 < rounds) {
#pragma unroll
            for (int i = 0; i < 4; i++) nxt[i] = src[(r + 1) * 4 + i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t w[4] = {cur[i].x, cur[i].y, cur[i].z, cur[i].w};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                uint32_t nl, acc;
                eng.step(st, (w[k >> 2] >> (8 * (k & 3))) & 0xffu, nl, acc);
                res.push(nl, acc);
            }
            if (__any(res.bits >> 15)) res.flush();      // <= 16 more results fit before the next check
        }
        res.flush();
#pragma unroll
        for (int i = 0; i < 4; i++) cur[i] = nxt[i];
    }
    pos += (size_t)rounds * kRound;

    // ---- tail of the corpus inside my stripe (only the last stripe has one), byte by byte
    for (; pos < my_end; pos++) {
        uint32_t nl, acc;
        eng.step(st, bytes[pos], nl, acc);
        res.push(nl, acc);
        if (res.bits >> 30) res.flush_lane();
    }
    res.flush_lane();

    // ---- follow my last line past the stripe end.  It is mine iff I started it: I began at a line start or
    // saw a '\n' inside my stripe, and my stripe does not end exactly on a '\n'.
    const bool started = fresh || res.line > first_line;
    if (started && bytes[my_end - 1] != '\n') {
        uint32_t nl = 0, acc = 0;
        for (; pos < nbytes && !nl; pos++) eng.step(st, bytes[pos], nl, acc);
        if (!nl) eng.step(st, '\n', nl, acc);       // the corpus ends without '\n': end of data ends the line
        res.push(nl, acc);
        res.flush_lane();
    }
}

// ============================================================================================ extents kernel
// One lane per item; bytes come straight from HBM/L2.  Used for explicit (offset,len) batches, for the
// iterator facade's single strings, and wherever '\n' is an ordinary character.
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
    for (size_t p = b; p < e; p++) {
        uint32_t c = bytes[p];
        if (c == 0 || c >= 0x80) { eng.kill(st); break; }
        eng.step(st, c);
    }
    accept[i] = eng.accepting(st) ? 1 : 0;
}

// ============================================================================================ line index
// counts[g] = number of '\n' in stripe g, streamed exactly like the match kernel streams it.
__global__ __launch_bounds__(kThreads) void count_newlines_kernel(const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                   uint32_t *__restrict__ counts, size_t nstripes) {
    const size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (g >= nstripes) return;
    const size_t start = g * (size_t)kStripe;
    const size_t end = start + kStripe < nbytes ? start + kStripe : nbytes;
    const uint4 *src = reinterpret_cast<const uint4 *>(bytes + start);
    const int units = (int)((end - start) / 16);
    uint32_t cnt = 0;
    int u = 0;
    for (; u + 4 <= units; u += 4) {
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
            }
        }
    }
    for (size_t p = start + (size_t)u * 16; p < end; p++) cnt += bytes[p] == '\n';
    counts[g] = cnt;
}

// exclusive scan of n counts into n+1 bases (single workgroup, chunked)
__global__ __launch_bounds__(1024) void scan_counts_kernel(const uint32_t *__restrict__ counts, uint64_t *__restrict__ base, size_t n) {
    __shared__ uint64_t sums[1024];
    const size_t chunk = (n + 1023) / 1024;
    const size_t lo = threadIdx.x * chunk < n ? threadIdx.x * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
    uint64_t s = 0;
    for (size_t i = lo; i < hi; i++) s += counts[i];
    sums[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; i++) { uint64_t v = sums[i]; sums[i] = run; run += v; }
        base[n] = run;
    }
    __syncthreads();
    uint64_t run = sums[threadIdx.x];
    for (size_t i = lo; i < hi; i++) { base[i] = run; run += counts[i]; }
}

template <class Engine, class Program>
int launch_stripes(const Program &p, size_t table_bytes, const uint8_t *bytes, size_t nbytes, const uint64_t *stripe_base,
                   size_t nstripes, uint8_t *accept, void *stream) {
    if (!nstripes) return 0;
    auto k = match_stripes_kernel<Engine, Program>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)table_bytes);
    if (e != hipSuccess) return (int)e;
    size_t blocks = (nstripes + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), table_bytes, (hipStream_t)stream, p, bytes, nbytes, stripe_base, accept);
    return (int)hipGetLastError();
}

template <class Engine, class Program>
int launch_extents(const Program &p, size_t table_bytes, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim,
                   uint8_t *accept, void *stream) {
    if (!nitems) return 0;
    auto k = match_extents_kernel<Engine, Program>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)table_bytes);
    if (e != hipSuccess) return (int)e;
    size_t blocks = (nitems + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), table_bytes, (hipStream_t)stream, p, bytes, off, nitems, trim, accept);
    return (int)hipGetLastError();
}

}  // namespace

int count_newlines_per_stripe(const uint8_t *bytes, size_t nbytes, uint32_t *counts, size_t nstripes, void *stream) {
    if (!nstripes) return 0;
    size_t blocks = (nstripes + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(count_newlines_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, bytes, nbytes, counts, nstripes);
    return (int)hipGetLastError();
}
int scan_counts(const uint32_t *counts, uint64_t *base, size_t n, void *stream) {
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, base, n);
    return (int)hipGetLastError();
}

#define RRX_NFA_DISPATCH(CALL)                                                      \
    switch (p.W) {                                                                  \
    case 1: return CALL(1);                                                         \
    case 2: return CALL(2);                                                         \
    case 3: return CALL(3);                                                         \
    case 4: return CALL(4);                                                         \
    case 5: case 6: return CALL(6);                                                 \
    case 7: case 8: return CALL(8);                                                 \
    default: return (int)hipErrorInvalidValue;                                      \
    }

// The device tables are padded to the instantiated width by the caller (NfaDevice::W is the padded width).
int match_stripes_nfa(const NfaDevice &p, const uint8_t *bytes, size_t nbytes, const uint64_t *stripe_base, size_t nstripes,
                      uint8_t *accept, void *stream) {
#define CALL(WW) launch_stripes<LineNfaEngine<WW>, NfaDevice>(p, LineNfaEngine<WW>::lds_bytes(p), bytes, nbytes, stripe_base, nstripes, accept, stream)
    RRX_NFA_DISPATCH(CALL)
#undef CALL
}
int match_stripes_dfa(const LineDfaDevice &p, const uint8_t *bytes, size_t nbytes, const uint64_t *stripe_base, size_t nstripes,
                      uint8_t *accept, void *stream) {
    if (p.wide) return launch_stripes<LineDfaEngine<true>, LineDfaDevice>(p, LineDfaEngine<true>::lds_bytes(p), bytes, nbytes, stripe_base, nstripes, accept, stream);
    return launch_stripes<LineDfaEngine<false>, LineDfaDevice>(p, LineDfaEngine<false>::lds_bytes(p), bytes, nbytes, stripe_base, nstripes, accept, stream);
}
int match_extents_nfa(const NfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                      void *stream) {
#define CALL(WW) launch_extents<PlainNfaEngine<WW>, NfaDevice>(p, PlainNfaEngine<WW>::lds_bytes(p), bytes, off, nitems, trim, accept, stream)
    RRX_NFA_DISPATCH(CALL)
#undef CALL
}
int match_extents_dfa(const DfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                      void *stream) {
    return launch_extents<PlainDfaEngine, DfaDevice>(p, PlainDfaEngine::lds_bytes(p), bytes, off, nitems, trim, accept, stream);
}

}  // namespace dev
}  // namespace rrx
