// kernels_common.hpp — shared device code of the kernel translation units (kernels_table.hip, kernels_nfa.hip,
// kernels_coop.hip).  Originally one file: hand-written CDNA4 (gfx950) kernels for the RoaringRegex hot path.
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
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

#include "device.hpp"

namespace rrx {
namespace dev {
namespace {

constexpr uint32_t kEndsOnNewline = 0x80000000u;         // bit 31 of counts[k]: stripe k ends on a '\n'
constexpr uint64_t kFreshStripe = 1ull << 63;             // bit 63 of stripe_base[k]: stripe k begins at the start of a line
__device__ __forceinline__ uint64_t line_of(uint64_t base) { return base & ~kFreshStripe; }

// Text loads are plain loads: non-temporal ones stop the 8 loads of a 128-byte line from merging into one request
// (measured -47 %, profiles/r01_v7_result_path_probes.txt).
__device__ __forceinline__ uint4 load_text(const uint4 *p) { return *p; }

// Workgroup copy of a device table into LDS, 16 bytes per lane per access and kBatch accesses in flight per lane
// (instead of one dword per trip of a `for (i = tid; i < n; i += blockDim.x)` loop).  Measured neutral on the stride-2
// engine, same box, old against new build (profiles/r02_headline_ab.txt): the table copy is not where a launch's fixed
// cost goes.  `add` goes onto every dword (tables whose entries hold LDS addresses).  src and dst are 16-byte aligned,
// nbytes is a multiple of 4.
template <int kBatch = 4>
__device__ __forceinline__ void copy_table_to_lds(void *dst_lds, const void *src, uint32_t nbytes, uint32_t add = 0) {
    uint4 *d = static_cast<uint4 *>(dst_lds);
    const uint4 *s = static_cast<const uint4 *>(src);
    const uint32_t nvec = nbytes / 16;
    for (uint32_t base = threadIdx.x; base < nvec; base += kBatch * blockDim.x) {
        uint4 v[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const uint32_t i = base + k * blockDim.x;
            if (i < nvec) v[k] = s[i];
        }
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const uint32_t i = base + k * blockDim.x;
            if (i < nvec) d[i] = make_uint4(v[k].x + add, v[k].y + add, v[k].z + add, v[k].w + add);
        }
    }
    const uint32_t tail = nvec * 4 + threadIdx.x;                 // the last 0..3 dwords
    if (tail < nbytes / 4) static_cast<uint32_t *>(dst_lds)[tail] = static_cast<const uint32_t *>(src)[tail] + add;
}

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
    template <class F> __device__ __forceinline__ void for_each_slot_mut(F &&f) { f(head); rest.for_each_slot_mut(f); }
};
template <>
struct TextRound<0> {
    __device__ __forceinline__ void load(const uint4 *) {}
    template <class F> __device__ __forceinline__ void for_each_slot(F &&) const {}
    template <class F> __device__ __forceinline__ void for_each_slot_mut(F &&) {}
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
    typename std::conditional<STAGED, uint32_t, uint64_t>::type word;      // STAGED: relative to the first word of the window (a
                                           //   workgroup's text holds far fewer than 2^37 lines: 32 bits, one register less)
    bool drop_first;
    uint32_t drop_mask = 1u;               // the result bits of one line (kernels that report two bits per line: 3)
    bool writer = true;                    // wave-cooperative kernels: every lane mirrors the bookkeeping, one writes
    uint32_t *__restrict__ out;            // STAGED: already advanced to the first word of the window
    uint32_t *stage = nullptr;
    uint32_t stage_words = kStageWords;    // words of the window

    __device__ __forceinline__ void begin(uint64_t first_line, bool drop, uint32_t *bitmap) {
        word = first_line >> 5; fill = (uint32_t)first_line & 31u; drop_first = drop; out = bitmap;
    }
    __device__ __forceinline__ void begin_staged(uint64_t first_line, uint64_t window_word, bool drop, uint32_t *bitmap, uint32_t *lds) {
        word = (decltype(word))((first_line >> 5) - window_word); fill = (uint32_t)first_line & 31u; drop_first = drop;
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
            if (drop_first) { rev &= ~drop_mask; drop_first = false; }
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
// packs its verdicts from bit 0 of its OWN stream and stores the stream word by word into the slab in HBM,
// slab[word][stripe]: word k of ALL stripes is one dense row (rows padded to whole workgroups); a lane of 4 KiB writes
// about three words.  (A slab per workgroup, slab[workgroup][word][lane], put those few rows 528 KiB apart; the dense
// rows measure 2-3 % better on average, within a process-to-process spread of +-5 % that both layouts show:
// profiles/r02_one_shot_breakdown.txt.)  The per-stripe newline counts the lanes
// write on the side are scanned afterwards, and compact_streams_kernel shifts every lane's stream to its place in the
// accept bitmap.  bit k of a lane's stream = the k-th line end it saw (the first one belongs to the lane before if the
// stripe starts inside a line: the compaction drops it, as ResultsT does with drop_first).
struct LocalResults {
    uint32_t bits = 1, outw = 0, fill = 0, seen = 0, k = 0;
    uint32_t *__restrict__ dst;            // &slab[0][my stripe]
    uint32_t row;                          // stripes per slab row
    __device__ __forceinline__ void begin(uint32_t *slab_lane, uint32_t row_stripes) { dst = slab_lane; row = row_stripes; }
    __device__ __forceinline__ void push(uint32_t nl, uint32_t acc) { bits = (bits << nl) | acc; }
    __device__ __forceinline__ void flush() {
        const int n = 31 - __clz((int)bits);
        if (n > 0) {
            const uint32_t rev = __brev(bits & ((1u << n) - 1u)) >> (32 - n);      // oldest line at bit 0
            outw |= rev << fill;
            uint32_t nf = fill + (uint32_t)n;
            if (nf >= 32u) {
                dst[(size_t)k * row] = outw;
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
        if (fill) dst[(size_t)k * row] = outw;
    }
};
constexpr uint32_t kCountMask = 0x3fffffffu;             // counts[g]: bits 0..29 = '\n' in stripe g
constexpr uint32_t kExtraResult = 0x40000000u;           // bit 30 (one-pass mode): the lane's stream holds one result more than
                                                         //   its stripe has '\n' (a line followed past the stripe end, or the
                                                         //   last line of a corpus that does not end in '\n')
__host__ __device__ inline size_t slab_words_per_lane(uint32_t stripe) { return stripe / 32 + 1; }

// ONEPASS (rrx_match_device): no line index yet - see LocalResults and dfa2_body.  The engines handle bytes >= 0x80 and 0x00
// themselves here (the clamped wide table, class tables, empty B rows).
template <class Engine, class Program, bool ONEPASS = false>
__device__ __forceinline__ void match_stripes_body(const Program &prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                   uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                   uint32_t *__restrict__ accept_bits, uint32_t stage_off, uint32_t stage_words,
                                                   uint32_t *__restrict__ counts = nullptr, uint32_t *__restrict__ slabs = nullptr) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr bool kWindow = Engine::kStaged && !ONEPASS;
    // result window of the workgroup (ResultsT<true>), behind the tables: their entries hold 16-bit LDS addresses
    uint32_t *stage = reinterpret_cast<uint32_t *>(smem + stage_off);
    Engine eng;
    eng.load(prog, smem);
    if (kWindow)
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
    typename std::conditional<ONEPASS, LocalResults, ResultsT<Engine::kStaged>>::type res;
    if constexpr (ONEPASS) {
        res.begin(slabs + g, gridDim.x * kThreads);
    } else {
        const uint64_t my_base = stripe_base[g];
        fresh = (my_base & kFreshStripe) != 0;
        if constexpr (Engine::kStaged) { res.begin_staged(line_of(my_base), window_word, !fresh, accept_bits, stage); res.stage_words = stage_words; }
        else res.begin(line_of(my_base), !fresh, accept_bits);
    }
    typename Engine::State st = fresh ? eng.fresh() : eng.skipping();

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
    const uint32_t newlines = res.seen;                              // '\n' inside my stripe

    // ---- follow my last line past the stripe end.  It is mine iff I started it: I began at a line start or
    // saw a '\n' inside my stripe, and my stripe does not end exactly on a '\n'.
    if (ONEPASS && res.seen == 0) fresh = g == 0 || bytes[start - 1] == '\n';   // a stripe without any '\n': whose line is it?
    const bool started = fresh || res.seen > 0;
    bool followed = false;
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
        followed = true;
    }
    res.finish();
    if (ONEPASS) counts[g] = newlines | (followed ? kExtraResult : 0u) | (bytes[my_end - 1] == '\n' ? kEndsOnNewline : 0u);
    }
    if (kWindow) {
        __syncthreads();                             // write the window out: consecutive lanes, consecutive words
        for (uint32_t i = threadIdx.x; i < stage_words; i += kThreads) {
            const uint32_t v = stage[i];
            if (v) atomicOr(&accept_bits[window_word + i], v);
        }
    }
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
__global__ __launch_bounds__(kThreads) void match_stripes_onepass_kernel(Program prog, const uint8_t *__restrict__ bytes, size_t nbytes, uint32_t stripe,
                                                                          uint32_t *__restrict__ counts, uint32_t *__restrict__ slabs) {
    match_stripes_body<Engine, Program, true>(prog, bytes, nbytes, stripe, nullptr, nullptr, 0, 0, counts, slabs);
}
template <class Engine, class Program>
__global__ __launch_bounds__(kThreads) void match_extents_kernel(Program prog, const uint8_t *__restrict__ bytes,
                                                                  const uint64_t *__restrict__ off, size_t nitems, uint32_t trim,
                                                                  uint8_t *__restrict__ accept, const uint32_t *__restrict__ only_if) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if (only_if && !*only_if) return;            // queued behind the stripe-wise kernel as its fallback: the batch was fit, nothing to do
    Engine eng;
    eng.load(prog, smem);
    __syncthreads();
    // (one item per lane when the grid covers the batch; the predicated fallback is launched with a bounded grid and strides)
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nitems; i += (size_t)gridDim.x * kThreads) {
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
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel, device and size increase, not once per launch.
// (One slot array per kernel: `slots` is a function-local static of the calling template instantiation.)
constexpr int kMaxDevices = 64;
struct LdsAttr { std::atomic<int> bytes[kMaxDevices]; std::atomic<int> at_zero_ok{0}; };
// at_zero: the kernel addresses a table by ABSOLUTE LDS address starting at 0 (B rows of the NFA lane engines, the search
// kernel's tables), which holds as long as it has no static LDS in front of its dynamic LDS.  Checked here, once per kernel,
// on the host - a launch that would break the assumption is refused (round 2 aborted the GPU process from inside the kernel).
inline hipError_t ensure_dynamic_lds(LdsAttr &slots, const void *kernel, size_t bytes, bool at_zero = false) {
    if (at_zero && !slots.at_zero_ok.load(std::memory_order_acquire)) {
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, kernel);
        if (e != hipSuccess) return e;
        if (fa.sharedSizeBytes != 0) return hipErrorInvalidConfiguration;
        slots.at_zero_ok.store(1, std::memory_order_release);
    }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (slots.bytes[dev].load(std::memory_order_acquire) >= (int)bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) slots.bytes[dev].store((int)bytes, std::memory_order_release);
    return e;
}
template <class Engine, class = void> struct lds_at_zero : std::false_type {};
template <class Engine> struct lds_at_zero<Engine, std::enable_if_t<Engine::kLdsAtZero>> : std::true_type {};

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
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(k), lds, lds_at_zero<Engine>::value);
    if (e != hipSuccess) return (int)e;
    size_t blocks = (nstripes + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), lds, (hipStream_t)stream, p, bytes, nbytes, stripe, stripe_base, accept, stage_off, stage_words);
    return (int)hipGetLastError();
}

template <class Engine, class Program>
int launch_extents(const Program &p, size_t table_bytes, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim,
                   uint8_t *accept, void *stream, const uint32_t *only_if = nullptr) {
    if (!nitems) return 0;
    auto k = match_extents_kernel<Engine, Program>;
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(k), table_bytes);
    if (e != hipSuccess) return (int)e;
    size_t blocks = (nitems + kThreads - 1) / kThreads;
    if (only_if && blocks > 1024) blocks = 1024;      // the fallback behind the stripe-wise kernel mostly has nothing to do: a grid that ends at once (86 k workgroups: 33 us)
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), table_bytes, (hipStream_t)stream, p, bytes, off, nitems, trim, accept, only_if);
    return (int)hipGetLastError();
}

template <class Engine, class Program>
int launch_onepass(const Program &p, size_t table_bytes, const uint8_t *bytes, size_t nbytes, uint32_t stripe, size_t nstripes, uint32_t *counts,
                   uint32_t *slabs, void *stream) {
    if (!nstripes) return 0;
    auto k = match_stripes_onepass_kernel<Engine, Program>;
    static LdsAttr attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(k), table_bytes, lds_at_zero<Engine>::value);
    if (e != hipSuccess) return (int)e;
    size_t blocks = (nstripes + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), table_bytes, (hipStream_t)stream, p, bytes, nbytes, stripe, counts, slabs);
    return (int)hipGetLastError();
}

}  // namespace
}  // namespace dev
}  // namespace rrx
