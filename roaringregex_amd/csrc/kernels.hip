// kernels.hip — hand-written CDNA4 (gfx950) kernels for the RoaringRegex hot path.
//
// Replaces, for a whole batch of '\n'-delimited strings at once:
//   AcceptanceIterator::operator++(int)   regex.h:156-159   (consume the string)
//   Processor::shift<true>                NFA.cc:72-102     (per-byte state-set transition)
//   Processor::operator*()                NFA.cc:103-107    (accepting?)
// Pure integer/bitwise work, HBM-read bound by design: no MFMA.
//
// Batch kernel (match_tiles<Engine>):
//   * one 256-thread workgroup per 32 KiB tile of text; the tile (+ a look-ahead halo) is staged into LDS
//     with coalesced 16 B/lane loads and an XOR swizzle of the 16-byte unit index, so that the per-lane
//     strided ds_read_b128 of "my 128-byte segment" is bank-conflict free;
//   * lane l scans the lines that START in its segment and follows the last one past the segment end
//     (into LDS halo, then HBM), so every line is stepped by exactly one lane from its first byte;
//   * the automaton tables live in LDS, the state set lives in registers;
//   * per 16-byte unit a lane ORs {newline mask, accept mask} into a positional LDS bitmap; after a
//     barrier the workgroup ranks the newline bits (popcount + block scan) and writes accept[line]
//     with line = tile_base[tile] + rank: no line-offset array is ever read.
#include <hip/hip_runtime.h>

#include "device.hpp"

namespace rrx {
namespace dev {
namespace {

constexpr int kUnits = (kTile + kHalo) / 16;        // staged 16-byte units
constexpr int kBitWords = kTile / 16;               // one u32 {nl:16, acc:16} per tile unit

__device__ __forceinline__ int swz(int unit) { return unit ^ ((unit >> 4) & 15); }

// ------------------------------------------------------------------------------------------ engines
template <int W>
struct NfaEngine {
    static constexpr int kWords = W;
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
    __device__ __forceinline__ void reset(State &st) const {
#pragma unroll
        for (int i = 0; i < W; i++) st.s[i] = m.init[i];
    }
    __device__ __forceinline__ void kill(State &st) const {
#pragma unroll
        for (int i = 0; i < W; i++) st.s[i] = 0;
    }
    __device__ __forceinline__ bool accepting(const State &st) const {
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < W; i++) a |= st.s[i] & m.fin[i];
        return a != 0;
    }
    // next = ( ((S << 1) & CHAIN) | (S & SELF) | OR_{e in S & EXC} X[e] ) & B[c]
    __device__ __forceinline__ void step(State &st, uint32_t c) const {
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

struct DfaEngine {
    struct State { uint32_t s; };
    const uint8_t *cls;     // LDS [256]
    const uint16_t *next;   // LDS [nstates][ncls]
    const uint8_t *acc;     // LDS [nstates]
    uint32_t ncls, start;

    static size_t lds_bytes(const DfaDevice &p) {
        size_t t = (size_t)p.nstates * p.ncls * 2;
        t = (t + 15) & ~(size_t)15;
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

// ------------------------------------------------------------------------------------------ batch kernel
template <class Engine, class Program>
__global__ __launch_bounds__(kThreads) void match_tiles_kernel(Program prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                const uint64_t *__restrict__ tile_base, size_t ntiles,
                                                                uint8_t *__restrict__ accept) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint4 *text = reinterpret_cast<uint4 *>(smem);                             // kUnits swizzled 16-byte units
    uint32_t *bits = reinterpret_cast<uint32_t *>(smem + (size_t)kUnits * 16); // kBitWords
    uint32_t *scan = bits + kBitWords;                                         // 8 words
    uint8_t *tables = reinterpret_cast<uint8_t *>(scan + 8);

    const size_t tile = blockIdx.x;
    const size_t tile_start = tile * (size_t)kTile;
    const int tid = threadIdx.x;

    Engine eng;
    eng.load(prog, tables);

    // ---- stage text: coalesced 16 B per lane, swizzled unit index
    const size_t avail = nbytes - tile_start;                                  // > 0 by construction of the grid
    for (int u = tid; u < kUnits; u += kThreads) {
        size_t off = (size_t)u * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (off + 16 <= avail) v = *reinterpret_cast<const uint4 *>(bytes + tile_start + off);
        else if (off < avail) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (size_t k = 0; off + k < avail; k++) w[k >> 2] |= (uint32_t)bytes[tile_start + off + k] << (8 * (k & 3));
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        text[swz(u)] = v;
    }
    for (int i = tid; i < kBitWords; i += kThreads) bits[i] = 0;
    const bool tile_mid_line = tile_start > 0 && bytes[tile_start - 1] != '\n';
    __syncthreads();

    // ---- scan my segment
    const size_t seg_start = tile_start + (size_t)tid * kSeg;
    if (seg_start < nbytes) {
        const size_t seg_end = seg_start + kSeg;
        bool active;
        if (tid == 0) active = !tile_mid_line;
        else {
            int prev = tid * kSeg - 1;
            const uint8_t *tb = reinterpret_cast<const uint8_t *>(&text[swz(prev >> 4)]);
            active = tb[prev & 15] == '\n';
        }
        typename Engine::State st;
        eng.reset(st);
        bool boundary = true;          // the next byte starts a line
        bool done = false;
        for (int u = tid * (kSeg / 16); !done; u++) {
            const size_t pos = tile_start + (size_t)u * 16;
            uint32_t w[4] = {0, 0, 0, 0};
            if (u < kUnits) {
                uint4 v = text[swz(u)];
                w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            } else if (pos + 16 <= nbytes) {
                uint4 v = *reinterpret_cast<const uint4 *>(bytes + pos);
                w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            } else {
                for (size_t k = 0; pos + k < nbytes; k++) w[k >> 2] |= (uint32_t)bytes[pos + k] << (8 * (k & 3));
            }
            uint32_t unit_bits = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (done) break;
                const size_t p = pos + k;
                bool is_end = p >= nbytes;
                uint32_t c = (w[k >> 2] >> (8 * (k & 3))) & 0xff;
                if (is_end) {
                    // end of corpus terminates an unfinished last line like a '\n' would
                    if (!(active && !boundary)) { done = true; break; }
                    c = '\n';
                }
                if (c == '\n') {
                    if (active) {
                        uint32_t a = eng.accepting(st) ? 0x10001u : 0x1u;
                        if (p < tile_start + kTile) unit_bits |= a << k;
                        else accept[tile_base[p / kTile]] = (uint8_t)(a >> 16);
                    }
                    active = true;
                    boundary = true;
                    eng.reset(st);
                    if (is_end || p + 1 >= seg_end) done = true;
                } else {
                    if (!active && p + 1 >= seg_end) { done = true; break; }
                    if (active) {
                        if (c == 0 || c >= 0x80) eng.kill(st); else eng.step(st, c);
                    }
                    boundary = false;
                }
            }
            if (unit_bits) atomicOr(&bits[u], unit_bits);       // u < kBitWords whenever unit_bits != 0
        }
    }
    __syncthreads();

    // ---- rank the newline bits of the tile and write accept[line]
    uint32_t mine[kSeg / 16];
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < kSeg / 16; i++) { mine[i] = bits[tid * (kSeg / 16) + i]; cnt += __popc(mine[i] & 0xffffu); }
    uint32_t incl = cnt;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) scan[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int i = 0; i < wave; i++) wave_off += scan[i];
    uint64_t line = tile_base[tile] + (tile_mid_line ? 1 : 0) + wave_off + (incl - cnt);
#pragma unroll
    for (int i = 0; i < kSeg / 16; i++) {
        uint32_t nl = mine[i] & 0xffffu, ac = mine[i] >> 16;
        while (nl) {
            int b = __ffs(nl) - 1;
            nl &= nl - 1;
            accept[line++] = (uint8_t)((ac >> b) & 1);
        }
    }
}

// ------------------------------------------------------------------------------------------ extents kernel
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

// ------------------------------------------------------------------------------------------ line index
__global__ __launch_bounds__(kThreads) void count_newlines_kernel(const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                   uint32_t *__restrict__ counts) {
    __shared__ uint32_t part[kThreads / 64];
    const size_t tile_start = (size_t)blockIdx.x * kTile;
    const size_t avail = nbytes - tile_start < (size_t)kTile ? nbytes - tile_start : (size_t)kTile;
    uint32_t cnt = 0;
    for (int u = threadIdx.x; u < kTile / 16; u += kThreads) {
        size_t o = (size_t)u * 16;
        if (o + 16 <= avail) {
            uint4 v = *reinterpret_cast<const uint4 *>(bytes + tile_start + o);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t x = w[j] ^ 0x0a0a0a0au;                          // zero byte <=> '\n'
                uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);   // exact zero-byte test
                cnt += __popc(z);
            }
        } else {
            for (size_t k = o; k < avail && k < o + 16; k++) cnt += bytes[tile_start + k] == '\n';
        }
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of ntiles counts into ntiles+1 bases (single workgroup, chunked)
__global__ __launch_bounds__(1024) void scan_counts_kernel(const uint32_t *__restrict__ counts, uint64_t *__restrict__ base, size_t n) {
    __shared__ uint64_t sums[1024];
    const size_t chunk = (n + 1023) / 1024;
    const size_t lo = threadIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
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
int launch_tiles(const Program &p, size_t table_bytes, const uint8_t *bytes, size_t nbytes, const uint64_t *tile_base, size_t ntiles,
                 uint8_t *accept, void *stream) {
    if (!ntiles) return 0;
    size_t lds = (size_t)kUnits * 16 + (size_t)kBitWords * 4 + 32 + table_bytes;
    auto k = match_tiles_kernel<Engine, Program>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(kThreads), lds, (hipStream_t)stream, p, bytes, nbytes, tile_base, ntiles, accept);
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

int count_newlines_per_tile(const uint8_t *bytes, size_t nbytes, uint32_t *tile_counts, size_t ntiles, void *stream) {
    if (!ntiles) return 0;
    hipLaunchKernelGGL(count_newlines_kernel, dim3((unsigned)ntiles), dim3(kThreads), 0, (hipStream_t)stream, bytes, nbytes, tile_counts);
    return (int)hipGetLastError();
}
int scan_tile_counts(const uint32_t *tile_counts, uint64_t *tile_base, size_t ntiles, void *stream) {
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, tile_counts, tile_base, ntiles);
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
int match_tiles_nfa(const NfaDevice &p, const uint8_t *bytes, size_t nbytes, const uint64_t *tile_base, size_t ntiles,
                    uint8_t *accept, void *stream) {
#define CALL(WW) launch_tiles<NfaEngine<WW>, NfaDevice>(p, NfaEngine<WW>::lds_bytes(p), bytes, nbytes, tile_base, ntiles, accept, stream)
    RRX_NFA_DISPATCH(CALL)
#undef CALL
}
int match_extents_nfa(const NfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                      void *stream) {
#define CALL(WW) launch_extents<NfaEngine<WW>, NfaDevice>(p, NfaEngine<WW>::lds_bytes(p), bytes, off, nitems, trim, accept, stream)
    RRX_NFA_DISPATCH(CALL)
#undef CALL
}
int match_tiles_dfa(const DfaDevice &p, const uint8_t *bytes, size_t nbytes, const uint64_t *tile_base, size_t ntiles,
                    uint8_t *accept, void *stream) {
    return launch_tiles<DfaEngine, DfaDevice>(p, DfaEngine::lds_bytes(p), bytes, nbytes, tile_base, ntiles, accept, stream);
}
int match_extents_dfa(const DfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim, uint8_t *accept,
                      void *stream) {
    return launch_extents<DfaEngine, DfaDevice>(p, DfaEngine::lds_bytes(p), bytes, off, nitems, trim, accept, stream);
}

}  // namespace dev
}  // namespace rrx
