// abi.cpp — the C ABI of librrx.so (include/rrx.h): host compile pipeline + device program upload + launches.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rrx.h"
#include "device.hpp"
#include "frontend.hpp"
#include "lower.hpp"

using namespace rrx;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) { g_err = msg; return code; }
int hip_fail(hipError_t e, const char *what) {
    return fail(RRX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(e_, #expr); } while (0)

constexpr uint32_t kMaxSubsetStates = 16384;   // subset construction is abandoned beyond this

struct DeviceTables {
    void *blob = nullptr;
    dev::NfaDevice nfa;
    dev::DfaDevice dfa;          // plain form (extents kernel)
    dev::LineDfaDevice line;     // line-mode form (batch kernel)
    dev::GroupNfaDevice group;   // group-cooperative NFA (16/32 lanes per string)
    dev::Dfa2Device dfa2;        // stride-2 line-mode table (corpora without bytes >= 0x80)
    dev::WaveNfaDevice block;    // wave-resident NFA (up to 65536 positions)
};

// The plain table in the wide line-table format with one more column: 0..127 byte values ('\n' an ordinary byte), 128 =
// any byte >= 0x80, 129 = END OF ITEM (verdict of the row, back to the start row).  For explicit items stepped stripe-wise.
struct ItemsTableOnDevice {
    void *blob = nullptr;
    dev::LineDfaDevice line;
};
struct Items2TableOnDevice {
    void *blob = nullptr;
    dev::Dfa2Device dfa2;
};
struct SearchTablesOnDevice {
    void *blob = nullptr;
    dev::SearchChunkDevice chunk;      // the stripe-wise kernel's tables (device.hpp)
};

// Small results a call has to hand back to the host (line totals, flags) are written by the call's last kernel into a slot
// of pinned, device-mapped host memory; the host then only waits for the stream.  (Round 2 read them with three
// hipMemcpyAsync into pageable stack variables and a synchronize: one call in twelve of the one-shot entry took 10.6 ms
// instead of 1.8 - BENCH_r02.json - with every kernel as fast as ever, profiles/r03_one_shot_calls.txt.)
struct Mailbox {
    volatile uint64_t *host = nullptr;
    uint64_t *dev = nullptr;
    int device = -1, slot = -1;
};
constexpr int kMailSlots = 64, kMailWords = 8;
struct MailPage { uint64_t *host = nullptr, *dev = nullptr; std::vector<int> free_slots; };
std::mutex g_mail_mu;
std::map<int, MailPage> g_mail;
int mailbox_acquire(int device, Mailbox *out) {
    std::lock_guard<std::mutex> lock(g_mail_mu);
    MailPage &pg = g_mail[device];
    if (!pg.host) {
        void *h = nullptr, *d = nullptr;
        hipError_t e = hipHostMalloc(&h, kMailSlots * kMailWords * sizeof(uint64_t), hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) { if (h) (void)hipHostFree(h); return hip_fail(e, "hipHostMalloc(mailbox)"); }
        pg.host = static_cast<uint64_t *>(h); pg.dev = static_cast<uint64_t *>(d);
        for (int i = kMailSlots - 1; i >= 0; i--) pg.free_slots.push_back(i);
    }
    if (pg.free_slots.empty()) return fail(RRX_ERR_HIP, "more than 64 synchronous calls in flight on one device");
    out->slot = pg.free_slots.back(); pg.free_slots.pop_back();
    out->device = device;
    out->host = pg.host + (size_t)out->slot * kMailWords;
    out->dev = pg.dev + (size_t)out->slot * kMailWords;
    return RRX_OK;
}
void mailbox_release(const Mailbox &m) {
    if (m.slot < 0) return;
    std::lock_guard<std::mutex> lock(g_mail_mu);
    g_mail[m.device].free_slots.push_back(m.slot);
}
// Returns the slot when the call leaves - but only once the device can no longer write it: a call that queued the mailing kernel
// and leaves before its stream has drained waits for the stream here, and a slot whose stream does not drain is never handed out again.
struct MailboxGuard {
    Mailbox m;
    hipStream_t stream = nullptr;
    bool queued = false;                                // the kernel that writes the slot has been launched on `stream`
    bool drained = false;                               // ... and the stream has been waited for since
    ~MailboxGuard() {
        if (queued && !drained && hipStreamSynchronize(stream) != hipSuccess) { (void)hipGetLastError(); return; }
        mailbox_release(m);
    }
};

int instantiated_width(uint32_t W) { return W <= 4 ? (int)W : W <= 6 ? 6 : W <= 8 ? 8 : W <= 12 ? 12 : 16; }

}  // namespace

struct rrx_regex {
    std::string pattern;
    RefAutomaton ref;
    Trimmed trimmed;
    bool has_nfa = false, has_dfa = false;
    NfaProgram nfa;
    NfaProgram nfa_wave;         // up to 4096 positions, no carry groups (wave-cooperative engine)
    bool has_wave = false;
    NfaProgram nfa_block;        // up to 65536 positions, exception edges in CSR form (wave-resident engine)
    bool has_block = false;
    DfaProgram dfa;
    Dfa2Program dfa2;
    bool has_dfa2 = false;
    // Order of the stride-2 table's rows and columns in LDS (empty: as numbered).  The order costs no memory and decides which
    // entries share an LDS bank: bank = (row slot * row words + column slot) mod 32.  State 0 (dead) keeps slot 0.
    std::vector<uint32_t> t2_row_slot, t2_col_slot;
    mutable TableOrderSearch t2_order;                   // the order search, in the background or in the caller of rrx_order_table
    mutable Dfa2OrderStats t2_order_stats;               // (under `mu`)
    std::atomic<int> opt_background_order{1};            // RRX_OPT_BACKGROUND_ORDER
    std::atomic<int> opt_search_anchored{1};             // RRX_OPT_SEARCH_ANCHORED
    std::atomic<int> opt_units_per_wg{0};                // RRX_OPT_UNITS_PER_WORKGROUP (0: one stripe per lane and launch)
    std::atomic<int> opt_sampled_table{1};               // RRX_OPT_SAMPLED_TABLE
    std::atomic<int> opt_flush_slots{0};                 // RRX_OPT_FLUSH_SLOTS (0: from the corpus' mean line length)
    // ---- the sampled table: AUTO ended on the NFA lane engine because the subset construction explodes; the sets a
    // text sample reaches are interned into a table with an ESCAPE state, the batch entry runs the stride-2 kernel on it (two result
    // bits per line) and lets the NFA engine decide the lines that escaped.  Built once: by the first rrx_match_corpus against a
    // corpus that carries a sample (in the background), or by rrx_learn_table (in the caller's thread).
    int requested_engine = 0;
    mutable OnceTask sampled_build;
    mutable std::atomic<bool> sampled_ready{false};      // (set under `mu` after the programs below are complete)
    mutable DfaProgram sampled_dfa;
    mutable Dfa2Program sampled_dfa2;
    mutable SampledTableStats sampled_stats;
    struct SampledOnDevice { void *blob = nullptr; dev::Dfa2Device d; };
    mutable std::map<int, const void *> sampled_counter;   // device -> where the last launch counted its escaped lines (under onepass_mu)
    // The same count, copied by every sampled launch into pinned host memory behind its kernels.
    // The NEXT launch looks at it without waiting (it shows the last launch that has finished): a corpus that escapes from the table
    // - not the text it was learnt from - retires the table (sampled_retired), the regex is back on the NFA engine at its own rate.
    mutable unsigned long long *h_sampled_seen = nullptr;  // hipHostMalloc, one slot per table generation (under onepass_mu)
    mutable unsigned long long sampled_prev_lines = 0;     // lines of the last sampled launch queued (under onepass_mu)
    mutable std::atomic<bool> sampled_retired{false};
    // (r4) A retired table is LEARNT AGAIN, from the sample of the corpus that retired it (the first one that carries a sample), up to
    // kSampledRelearns times: the build runs like the first one (beside the caller unless RRX_OPT_BACKGROUND_ORDER is 0), the regex stays
    // on the NFA engine meanwhile, and the new table is subject to the same two guards.  The old device tables are kept until rrx_free
    // (a launch queued on them may still be running).
    static constexpr uint32_t kSampledRelearns = 3;
    mutable uint32_t sampled_gen = 0;                      // generation of the table in use (under onepass_mu; slot of h_sampled_seen)
    mutable std::vector<std::unique_ptr<OnceTask>> sampled_relearn;      // (under onepass_mu)
    mutable std::vector<std::pair<int, void *>> sampled_old_blobs;       // (under mu)
    mutable std::map<int, SampledOnDevice> sampled_on_device;
    bool sampled_eligible() const { return requested_engine == RRX_ENGINE_AUTO && engine == RRX_ENGINE_NFA && !has_dfa && has_nfa; }
    // pieces x piece_bytes of text -> the table; false: nothing usable came out (the engine stays as it is)
    bool build_sampled(const uint8_t *text, uint32_t pieces, uint32_t piece_bytes, bool replace = false) const {
        const Reduced red = reduce(trimmed);
        DfaProgram d;
        Dfa2Program d2;
        SampledTableStats st;
        bool ok = false;
        for (uint32_t budget = 2048; budget >= 64 && !ok; budget /= 2) {       // the largest table whose stride-2 form fits the LDS
            if (!lower_dfa_sampled(red, text, pieces, piece_bytes, budget, d, &st)) return false;
            ok = d.nstates <= 4096 && lower_dfa2(d, 1024, d2) && (size_t)d2.nstates * (d2.ncols | 1u) * 4 <= dev::kDfa2MaxTable;
        }
        if (!ok || d.escaped.empty()) return false;      // (no escape state: the closure closed the table - lower_dfa would have too)
        // A table its own sample escapes from is the wrong tool: every escaped line is read a second time by the NFA engine, so
        // text whose live sets are NOT few (random a/b lines under (a|b)*a(a|b){40}: every line escapes) would run at a fraction of
        // the plain NFA engine's rate.  More than 2 % of the sample's lines: the engine stays as it is.
        if (st.sample_escapes * 50 > st.sample_lines) return false;
        if (replace) {
            // no sampled launch is being queued while the programs change (onepass_mu, taken before mu as match_corpus_sampled does)
            std::lock_guard<std::mutex> launches(onepass_mu);
            std::lock_guard<std::mutex> lock(mu);
            for (auto &kv : sampled_on_device) if (kv.second.blob) sampled_old_blobs.emplace_back(kv.first, kv.second.blob);
            sampled_on_device.clear();
            sampled_dfa = std::move(d); sampled_dfa2 = std::move(d2); sampled_stats = st;
            sampled_gen++;                               // (its own slot of h_sampled_seen: a late count of the old table's launches does not reach it)
            sampled_prev_lines = 0;
            sampled_retired.store(false);
            return true;
        }
        std::lock_guard<std::mutex> lock(mu);
        sampled_dfa = std::move(d); sampled_dfa2 = std::move(d2); sampled_stats = st;
        sampled_ready.store(true, std::memory_order_release);
        return true;
    }
    // the table is retired and `c` carries a text sample: learn it again from that (once per retirement, kSampledRelearns times in all)
    void relearn_sampled(const uint8_t *sample, uint32_t pieces, uint32_t piece_bytes) const {
        OnceTask *task = nullptr;
        {
            std::lock_guard<std::mutex> launches(onepass_mu);
            if (!sampled_retired.load() || sampled_relearn.size() >= kSampledRelearns) return;
            if (!sampled_relearn.empty() && sampled_relearn.back()->state() != OnceTask::kDone) return;      // (one at a time)
            sampled_relearn.emplace_back(new OnceTask());
            task = sampled_relearn.back().get();
        }
        // (started outside the lock: with RRX_OPT_BACKGROUND_ORDER 0 the job runs right here, and it takes onepass_mu itself to swap the table in)
        auto text = std::make_shared<std::vector<uint8_t>>(sample, sample + (size_t)pieces * piece_bytes);
        (void)task->start([this, text, pieces, piece_bytes]() { (void)build_sampled(text->data(), pieces, piece_bytes, /*replace=*/true); },
                          /*background=*/opt_background_order.load() != 0);
    }
    int sampled_tables(int device, dev::Dfa2Device *out) const {
        std::lock_guard<std::mutex> lock(mu);
        auto it = sampled_on_device.find(device);
        if (it != sampled_on_device.end()) { *out = it->second.d; return RRX_OK; }
        HIP_TRY(hipSetDevice(device));
        std::vector<uint32_t> T2;
        std::vector<uint16_t> P;
        SampledOnDevice t;
        build_dfa2_arrays_of(sampled_dfa2, {}, {}, T2, P, t.d);
        const size_t pb = (P.size() * 2 + 15) & ~(size_t)15;
        HIP_TRY(hipMalloc(&t.blob, pb + T2.size() * 4 + 16));
        hipError_t e = hipMemcpy(t.blob, P.data(), P.size() * 2, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(static_cast<uint8_t *>(t.blob) + pb, T2.data(), T2.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(t.blob); return hip_fail(e, "sampled table upload"); }
        t.d.P = static_cast<const uint16_t *>(t.blob);
        t.d.T2 = reinterpret_cast<const uint32_t *>(static_cast<uint8_t *>(t.blob) + pb);
        sampled_on_device.emplace(device, t);
        *out = t.d;
        return RRX_OK;
    }
    mutable std::vector<std::pair<int, void *>> t2_extra_blobs;      // tables uploaded again in the profiled order (device, blob)
    // The stride-2 tables as they go to the device, in the current order: T2 rows of `ncols | 1` entries (odd), R interleaved
    // copies, entry = LDS byte offset of the next row | lines << 16 | verdicts << 24; P = pair -> byte offset of its column.
    // `rows` / `cols`: an order to build for (else the current one: call with `mu` held)
    void build_dfa2_arrays(std::vector<uint32_t> &T2, std::vector<uint16_t> &P, dev::Dfa2Device &d, const std::vector<uint32_t> *rows = nullptr,
                           const std::vector<uint32_t> *cols = nullptr) const {
        build_dfa2_arrays_of(dfa2, rows ? *rows : t2_row_slot, cols ? *cols : t2_col_slot, T2, P, d);
    }
    static void build_dfa2_arrays_of(const Dfa2Program &dfa2, const std::vector<uint32_t> &rs, const std::vector<uint32_t> &cs, std::vector<uint32_t> &T2,
                                     std::vector<uint16_t> &P, dev::Dfa2Device &d) {
        const uint32_t D2 = dfa2.nstates, C2 = dfa2.ncols;
        const uint32_t s2 = C2 | 1u;
        uint32_t rep2 = 0;
        while (rep2 < 5 && (size_t)D2 * s2 * 4 * (2u << rep2) <= dev::kDfa2TableBudget && (size_t)C2 * 4 * (2u << rep2) <= 65535) rep2++;
        const uint32_t R2 = 1u << rep2;
        T2.assign((size_t)D2 * s2 * R2, 0);
        const bool ordered = rs.size() == D2 && cs.size() == C2 && rs[0] == 0;
        auto row_slot = [&](uint32_t st) { return ordered ? rs[st] : st; };
        auto col_slot = [&](uint32_t col) { return ordered ? cs[col] : col; };
        for (uint32_t st = 0; st < D2; st++)
            for (uint32_t col = 0; col < C2; col++) {
                const uint32_t v = dfa2.next2[(size_t)st * C2 + col];
                const uint32_t row_off = row_slot(v & 0xffffu) * s2 * 4 * R2;
                for (uint32_t k = 0; k < R2; k++) T2[((size_t)row_slot(st) * s2 + col_slot(col)) * R2 + k] = (row_off + 4 * k) | (v & 0xffff0000u);
            }
        const unsigned dim = dfa2.pair_dim;                 // 128; items form: 129 (code 128 = END OF ITEM: a row more, and the pad column 128)
        P.assign(dim * dev::kDfa2PStride, 0);
        for (unsigned c1 = 0; c1 < dim; c1++)
            for (unsigned c2 = 0; c2 < dim; c2++) P[c1 * dev::kDfa2PStride + c2] = (uint16_t)(col_slot(dfa2.pair_col[c1 * dim + c2]) * 4 * R2);
        d.nrows = D2; d.stride = s2 * R2; d.start_off = row_slot(dfa2.start) * s2 * 4 * R2; d.rep_log2 = rep2;
    }
    bool t2_order_applies() const {                      // single-copy tables only: interleaved copies already keep lanes apart
        return has_dfa2 && (size_t)dfa2.nstates * (dfa2.ncols | 1u) * 4 * 2 > dev::kDfa2TableBudget;
    }
    // What happens to a found order (runs in the searching thread).  For devices whose tables are already up the stride-2 arrays
    // are built and uploaded again WITHOUT `mu` - launches go on meanwhile on the table as numbered; `mu` is taken twice, briefly:
    // to read which devices are up, and to swap the slot vectors and the descriptors.  A device that comes up in between gets the
    // numbered order and keeps it (its own arrays agree with each other; results never depend on the order).
    void apply_t2_order(std::vector<uint32_t> &&rows, std::vector<uint32_t> &&cols, const Dfa2OrderStats &st) const {
        std::vector<std::pair<int, dev::Dfa2Device>> up;
        {
            std::lock_guard<std::mutex> lock(mu);
            for (auto &kv : on_device) up.emplace_back(kv.first, kv.second.dfa2);
        }
        struct Uploaded { int device; void *blob; dev::Dfa2Device d; };
        std::vector<Uploaded> done;
        for (auto &kv : up) {
            std::vector<uint32_t> T2;
            std::vector<uint16_t> P;
            dev::Dfa2Device d = kv.second;
            build_dfa2_arrays(T2, P, d, &rows, &cols);
            const size_t pb = (P.size() * 2 + 15) & ~(size_t)15;
            void *blob = nullptr;
            hipStream_t st2 = nullptr;
            bool ok = hipSetDevice(kv.first) == hipSuccess && hipMalloc(&blob, pb + T2.size() * 4) == hipSuccess &&
                      hipStreamCreateWithFlags(&st2, hipStreamNonBlocking) == hipSuccess;
            if (ok) ok = hipMemcpyAsync(blob, P.data(), P.size() * 2, hipMemcpyHostToDevice, st2) == hipSuccess &&
                         hipMemcpyAsync(static_cast<uint8_t *>(blob) + pb, T2.data(), T2.size() * 4, hipMemcpyHostToDevice, st2) == hipSuccess &&
                         hipStreamSynchronize(st2) == hipSuccess;
            if (st2) (void)hipStreamDestroy(st2);
            if (!ok) { (void)hipGetLastError(); if (blob) (void)hipFree(blob); continue; }       // (that device keeps the numbered order)
            d.P = static_cast<const uint16_t *>(blob);
            d.T2 = reinterpret_cast<const uint32_t *>(static_cast<uint8_t *>(blob) + pb);
            done.push_back(Uploaded{kv.first, blob, d});
        }
        std::lock_guard<std::mutex> lock(mu);
        rrx_regex *self = const_cast<rrx_regex *>(this);
        self->t2_row_slot.swap(rows); self->t2_col_slot.swap(cols);
        t2_order_stats = st;
        for (const Uploaded &u : done) {
            t2_extra_blobs.emplace_back(u.device, u.blob);
            auto it = on_device.find(u.device);
            if (it != on_device.end()) it->second.dfa2 = u.d;
        }
    }
    // First match against a corpus that carries a text sample: start the search in the background (the match itself, and the
    // next ones, run on the table as numbered until the new order is in) - unless the caller has forbidden library threads
    // (RRX_OPT_BACKGROUND_ORDER 0): then nothing happens here and rrx_order_table is the only way to an ordered table.
    // `now`: run it in the caller's thread (rrx_order_table).  Returns false if the order had been decided before.
    bool decide_t2_order(const uint8_t *sample, uint32_t lanes, uint32_t bytes_per_lane, bool now) const {
        if (t2_order.decided()) return false;
        if (!t2_order_applies() || !sample || lanes < 32) return t2_order.skip();
        if (!now && !opt_background_order.load()) return true;                   // (left undecided: rrx_order_table may still come)
        std::vector<uint8_t> copy(sample, sample + (size_t)lanes * bytes_per_lane);
        return t2_order.start(dfa2, std::move(copy), lanes, bytes_per_lane, /*background=*/!now,
                              [this](std::vector<uint32_t> &&r, std::vector<uint32_t> &&c, const Dfa2OrderStats &st) { apply_t2_order(std::move(r), std::move(c), st); });
    }
    dev::Dfa2Device dfa2_device(const DeviceTables *t) const { std::lock_guard<std::mutex> lock(mu); return t->dfa2; }
    int engine = 0;
    bool line_wide = false;      // DFA engine: byte-indexed rows (<= kWideMaxStates states) or class-indexed rows
    bool line_global = false;    // DFA engine: class-indexed table too large for LDS, kept in global memory
    mutable std::mutex mu;
    mutable std::map<int, DeviceTables> on_device;
    // search (built on first use): the forward "anything, then the pattern" DFA and the reverse DFA
    mutable int search_state = 0;        // 0 = not built, 1 = built, -1 = does not fit
    mutable DfaProgram search_fwd, search_rev;
    mutable SearchLineProgram search_line;  // stripe-wise form (nrows = 0: not built)
    mutable SearchLine2Program search_line2;    // its stride-2 form, what the stripe-wise kernel runs (nrows = 0: not built)
    mutable dev::SearchChunkDevice chunk_proto; // its layout on the device, without the pointers (nrows = 0: the line-per-lane kernels)
    mutable bool search_nullable = false;       // the pattern accepts the empty string: every offset is a match, no table (empty_matches)
    mutable std::map<int, SearchTablesOnDevice> search_on_device;
    mutable std::map<int, ItemsTableOnDevice> items_on_device;
    mutable std::map<int, Items2TableOnDevice> items2_on_device;
    mutable Dfa2Program items2_prog;                   // lowered at the first batch of items with separators
    mutable int items2_state = 0;                      // 0 not tried, 1 there, 2 does not fit
    std::atomic<int> items_stride2{1};                 // RRX_OPT_ITEMS_STRIDE2 (0: the byte-stride items kernel for trim 1 as well)
    // Scratch of the single-string entries (rrx_match_string / rrx_match_cstr): one grow-only device buffer per device,
    // kept across calls (a hipMalloc + hipFree pair per string cost more than the match itself).  `scratch_mu` is held
    // for the whole call: those entries are synchronous, concurrent callers of one regex take turns.
    struct Scratch { void *p = nullptr; size_t cap = 0; };
    mutable std::mutex scratch_mu;
    mutable std::map<int, Scratch> scratch;
    int scratch_for(int device, size_t bytes, void **out) const {      // call with `scratch_mu` held
        Scratch &sc = scratch[device];
        if (sc.cap < bytes) {
            if (sc.p) { (void)hipFree(sc.p); sc.p = nullptr; sc.cap = 0; }
            const size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
            hipError_t e = hipMalloc(&sc.p, want);
            if (e != hipSuccess) { sc.p = nullptr; return hip_fail(e, "hipMalloc(single-string scratch)"); }
            sc.cap = want;
        }
        *out = sc.p;
        return RRX_OK;
    }

    // Scratch of the one-shot entry (rrx_match_device: per-stripe counts, their scan and the lanes' verdict streams) and of
    // one-call explicit items (rrx_match_extents: the item index).  One grow-only buffer per device, kept until rrx_free.
    // Users on different streams are ordered on the DEVICE by an event recorded after each use (the host never waits):
    // onepass_for(..., stream) makes `stream` wait for the last user, onepass_done(stream) marks the new last use; both under
    // `onepass_mu`, held from the one to the other.
    struct EventScratch { void *p = nullptr; size_t cap = 0; hipEvent_t last = nullptr; bool used = false; };
    mutable std::mutex onepass_mu;
    mutable std::map<int, EventScratch> onepass_scratch;
    int onepass_for(int device, size_t bytes, void **out, hipStream_t stream) const {      // call with `onepass_mu` held
        EventScratch &sc = onepass_scratch[device];
        if (!sc.last) {
            hipError_t e = hipEventCreateWithFlags(&sc.last, hipEventDisableTiming);
            if (e != hipSuccess) { sc.last = nullptr; return hip_fail(e, "hipEventCreate(scratch)"); }
        }
        if (sc.cap < bytes) {
            if (sc.p) { (void)hipEventSynchronize(sc.last); (void)hipFree(sc.p); sc.p = nullptr; sc.cap = 0; sc.used = false; }
            hipError_t e = hipMalloc(&sc.p, bytes);
            if (e != hipSuccess) { sc.p = nullptr; return hip_fail(e, "hipMalloc(one-pass scratch)"); }
            sc.cap = bytes;
        }
        if (sc.used) {
            hipError_t e = hipStreamWaitEvent(stream, sc.last, 0);
            if (e != hipSuccess) return hip_fail(e, "hipStreamWaitEvent(scratch)");
        }
        *out = sc.p;
        return RRX_OK;
    }
    int onepass_done(int device, hipStream_t stream) const {                               // call with `onepass_mu` held
        EventScratch &sc = onepass_scratch[device];
        hipError_t e = hipEventRecord(sc.last, stream);
        if (e != hipSuccess) return hip_fail(e, "hipEventRecord(scratch)");
        sc.used = true;
        return RRX_OK;
    }

    ~rrx_regex() {
        t2_order.wait();
        sampled_build.wait();
        for (auto &task : sampled_relearn) task->wait();
        if (h_sampled_seen) {                            // (a copy into it may still be queued on the devices that ran the sampled table)
            for (auto &kv : sampled_on_device) { (void)hipSetDevice(kv.first); (void)hipDeviceSynchronize(); }
            for (auto &kv : sampled_old_blobs) { (void)hipSetDevice(kv.first); (void)hipDeviceSynchronize(); }
            (void)hipHostFree(h_sampled_seen);
        }
        for (auto &kv : sampled_on_device) if (kv.second.blob) { (void)hipSetDevice(kv.first); (void)hipFree(kv.second.blob); }
        for (auto &kv : sampled_old_blobs) { (void)hipSetDevice(kv.first); (void)hipFree(kv.second); }
        for (auto &kv : t2_extra_blobs) { (void)hipSetDevice(kv.first); (void)hipFree(kv.second); }
        for (auto &kv : scratch) if (kv.second.p) { (void)hipSetDevice(kv.first); (void)hipFree(kv.second.p); }
        for (auto &kv : onepass_scratch) {
            (void)hipSetDevice(kv.first);
            if (kv.second.last) { (void)hipEventSynchronize(kv.second.last); (void)hipEventDestroy(kv.second.last); }
            if (kv.second.p) (void)hipFree(kv.second.p);
        }
        for (auto &kv : on_device) if (kv.second.blob) { (void)hipSetDevice(kv.first); (void)hipFree(kv.second.blob); }
        for (auto &kv : search_on_device) if (kv.second.blob) { (void)hipSetDevice(kv.first); (void)hipFree(kv.second.blob); }
        for (auto &kv : items_on_device) if (kv.second.blob) { (void)hipSetDevice(kv.first); (void)hipFree(kv.second.blob); }
        for (auto &kv : items2_on_device) if (kv.second.blob) { (void)hipSetDevice(kv.first); (void)hipFree(kv.second.blob); }
    }
    // The stride-2 table of explicit items with a separator byte each (trim 1; lower_dfa2's items form): nullptr where the regex
    // has no stride-2 table or the items form - one symbol more - does not fit the same LDS region.
    bool items2_program_locked() const {                 // host side (call with `mu` held)
        if (has_dfa2 && items2_state == 0) {
            items2_state = lower_dfa2(dfa, 1024, items2_prog, /*items=*/true) &&
                           (size_t)items2_prog.nstates * (items2_prog.ncols | 1u) * 4 <= dev::kDfa2MaxTable ? 1 : 2;
        }
        return items2_state == 1;
    }
    bool items2_program() const { std::lock_guard<std::mutex> lock(mu); return items2_program_locked(); }
    const dev::Dfa2Device *items2_table(int device) const {
        std::lock_guard<std::mutex> lock(mu);
        auto it = items2_on_device.find(device);
        if (it != items2_on_device.end()) return it->second.blob ? &it->second.dfa2 : nullptr;
        Items2TableOnDevice t;
        if (items2_program_locked() && hipSetDevice(device) == hipSuccess) {
            std::vector<uint32_t> T2;
            std::vector<uint16_t> P;
            build_dfa2_arrays_of(items2_prog, std::vector<uint32_t>(), std::vector<uint32_t>(), T2, P, t.dfa2);
            const size_t pb = (P.size() * 2 + 15) & ~(size_t)15;
            if (pb <= dev::kDfa2PItemsBytes && hipMalloc(&t.blob, dev::kDfa2PItemsBytes + T2.size() * 4) == hipSuccess) {
                if (hipMemset(t.blob, 0, dev::kDfa2PItemsBytes) != hipSuccess ||
                    hipMemcpy(t.blob, P.data(), P.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(static_cast<uint8_t *>(t.blob) + dev::kDfa2PItemsBytes, T2.data(), T2.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
                    (void)hipFree(t.blob); t.blob = nullptr;
                }
            }
            t.dfa2.P = static_cast<const uint16_t *>(t.blob);
            t.dfa2.T2 = t.blob ? reinterpret_cast<const uint32_t *>(static_cast<uint8_t *>(t.blob) + dev::kDfa2PItemsBytes) : nullptr;
        }
        auto ins = items2_on_device.emplace(device, t);
        return ins.first->second.blob ? &ins.first->second.dfa2 : nullptr;
    }
    // nullptr: the plain table has too many states for 16-bit row offsets (or there is none)
    const dev::LineDfaDevice *items_table(int device) const {
        std::lock_guard<std::mutex> lock(mu);
        auto it = items_on_device.find(device);
        if (it != items_on_device.end()) return it->second.blob ? &it->second.line : nullptr;
        ItemsTableOnDevice t;
        // rows of kItemColumns entries (odd: a column's entries of different rows spread over all LDS banks), R interleaved
        // copies like the wide line table (lane l reads copy l % R: only banks congruent to l mod R)
        const uint32_t D = dfa.nstates, stride = dev::kItemColumns;
        uint32_t rep = 0;
        while (rep < 3 && (size_t)D * stride * 4 * (2u << rep) <= 60 * 1024) rep++;
        const uint32_t R = 1u << rep, row_bytes = stride * 4 * R;
        if (D && (size_t)D * row_bytes <= 65535 && hipSetDevice(device) == hipSuccess) {
            std::vector<uint32_t> T((size_t)D * stride * R, 0);
            for (uint32_t q = 0; q < D; q++)
                for (uint32_t c = 0; c < stride; c++) {
                    uint32_t v;
                    if (c <= 128) {                                                      // byte 2, bit 7: the row it leads to is accepting (what a trim-0
                        const uint32_t nx = dfa.next[(size_t)q * dfa.ncls + dfa.cls[c]];  // item that ends on this byte reports; as a shift count it is 0)
                        v = nx * row_bytes | (dfa.accepting[nx] ? 0x80u << 16 : 0u);
                    }
                    else if (c == dev::kItemEndColumn) v = dfa.start * row_bytes | 1u << 16 | (dfa.accepting[q] ? 1u << 24 : 0u);
                    else v = 0;                                                          // padding column: never read
                    for (uint32_t k = 0; k < R; k++) T[((size_t)q * stride + c) * R + k] = (v & 0xffff0000u) | ((v & 0xffffu) + 4 * k);
                }
            if (hipMalloc(&t.blob, T.size() * 4 + 16) == hipSuccess) {
                if (hipMemcpy(t.blob, T.data(), T.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(t.blob); t.blob = nullptr; }
            }
            t.line.nrows = D; t.line.stride = stride * R; t.line.start_off = dfa.start * row_bytes; t.line.wide = 1; t.line.rep_log2 = rep; t.line.in_global = 0;
            t.line.table = static_cast<const uint32_t *>(t.blob);
        }
        auto ins = items_on_device.emplace(device, t);
        return ins.first->second.blob ? &ins.first->second.line : nullptr;
    }

    // Host side of the search tables (call with `mu` held).  RRX_OK also for a pattern that accepts the empty string: it needs
    // no table (search_nullable), search_fwd / search_rev are built all the same (rrx_program_words).
    int build_search() const {
        if (search_state == 0) {
            const Reduced red = reduce(trimmed);
            const bool ok = search_dfas(red, kMaxSubsetStates, search_fwd, search_rev);
            search_nullable = rrx_accepts_empty(this) != 0;
            search_line = SearchLineProgram();
            search_line2 = SearchLine2Program();
            chunk_proto = dev::SearchChunkDevice();
            if (ok && !search_nullable) {
                // the product with the anchored table tells the hits whose match starts at the line start (no walk back); a
                // product beyond the row budget: the forward table alone (every hit walks)
                DfaProgram anchored;
                if (!(opt_search_anchored.load() && lower_dfa(red, kMaxSubsetStates, anchored) && lower_search_line(search_fwd, &anchored, 65534, search_line)) &&
                    !lower_search_line(search_fwd, nullptr, 65534, search_line))
                    search_line = SearchLineProgram();
            }
            if (search_line.nrows) {
                uint32_t column[256];
                for (int c = 0; c < 256; c++) column[c] = c == '\n' ? search_line.ncols - 1 : search_fwd.cls[c];
                if (!lower_search_line2(search_line, column, 16383, search_line2)) search_line2 = SearchLine2Program();
            }
            // the stripe-wise kernel's layout of that table: LDS if it fits beside the reverse table, the job pools and a result
            // window, else HBM/L2 (device.hpp: SearchChunkDevice)
            const SearchLine2Program &s2 = search_line2;
            if (s2.nrows && search_fwd.ncls < 128) {
                dev::SearchChunkDevice c;
                c.nrows = s2.nrows; c.ncols2 = s2.ncols; c.start_row = s2.start; c.skip_row = s2.skip;
                c.nr = search_rev.nstates; c.ncls = search_fwd.ncls; c.start_r = search_rev.start;
                uint32_t rb = (2 * s2.ncols + 3) & ~3u;
                if (((rb >> 2) & 1u) == 0) rb += 4;                           // an odd number of dwords per row: rows spread over the LDS banks
                c.row_bytes = rb; c.base_row = (dev::kSearchP8Bytes + rb - 1) / rb; c.in_global = 0;
                bool fits = s2.ncols <= 127 && c.base_row + s2.nrows <= 4096 && dev::search_chunks_lds_bytes(c) <= dev::kSearchChunkLdsBudget;
                if (!fits) {
                    c.row_bytes = 0; c.base_row = 0; c.in_global = 1;
                    fits = dev::search_chunks_lds_bytes(c) <= dev::kSearchChunkLdsBudget;      // (the reverse table has no global form)
                }
                if (fits) chunk_proto = c;
            }
            search_state = ok && (search_nullable || chunk_proto.nrows) ? 1 : -1;
        }
        return search_state == 1 ? RRX_OK
                                 : fail(RRX_ERR_UNSUPPORTED, "search tables too large for the device (the reverse DFA must fit 64 KiB of LDS, the forward "
                                                              "product table 65534 rows and 256 MiB)");
    }
    // The stripe-wise kernel's tables on `device` (uploaded once); *out = nullptr for a pattern that accepts the empty string.
    int search_tables(int device, const dev::SearchChunkDevice **out) const {
        std::lock_guard<std::mutex> lock(mu);
        int rc = build_search();
        if (rc) return rc;
        *out = nullptr;
        if (search_nullable) return RRX_OK;
        auto it = search_on_device.find(device);
        if (it != search_on_device.end()) { *out = &it->second.chunk; return RRX_OK; }
        HIP_TRY(hipSetDevice(device));
        std::vector<uint8_t> host;
        auto put = [&](const void *p, size_t n) { size_t off = (host.size() + 15) & ~(size_t)15; host.resize(off + n); std::memcpy(host.data() + off, p, n); return off; };
        const size_t oC = put(search_fwd.cls, 256);
        SearchTablesOnDevice t;
        // The line-mode product table in its stride-2 form (lower_search_line2), laid out for LDS (16-bit entries, a byte-wide
        // pair table) or for HBM/L2 (32-bit entries, a 16-bit pair table in LDS): device.hpp, SearchChunkDevice.
        size_t oP = 0, oT = 0, oTA = 0, oRV = 0;
        const uint32_t K = search_fwd.ncls, NR = search_rev.nstates;
        const SearchLine2Program &s2 = search_line2;
        t.chunk = chunk_proto;
        if (!t.chunk.in_global) {
            const uint32_t rb = t.chunk.row_bytes, br = t.chunk.base_row;
            std::vector<uint8_t> p8(dev::kSearchP8Bytes, 0);
            for (unsigned c1 = 0; c1 < 128; c1++)
                for (unsigned c2 = 0; c2 < 128; c2++) p8[c1 * dev::kSearchP8Stride + c2] = (uint8_t)(2 * s2.pair_col[c1 * 128 + c2]);
            auto lay = [&](const std::vector<uint32_t> &src) {
                std::vector<uint16_t> T((size_t)s2.nrows * (rb / 2), 0);
                for (uint32_t r = 0; r < s2.nrows; r++)
                    for (uint32_t c = 0; c < s2.ncols; c++) {
                        const uint32_t v = src[(size_t)r * s2.ncols + c];
                        T[(size_t)r * (rb / 2) + c] = (uint16_t)((br + (v & 0xffffffu)) << 4 | (v >> 24));
                    }
                return T;
            };
            const std::vector<uint16_t> T = lay(s2.first), TA = lay(s2.all);
            oP = put(p8.data(), p8.size());
            oT = put(T.data(), T.size() * 2);
            oTA = put(TA.data(), TA.size() * 2);
        } else {
            std::vector<uint16_t> p16((size_t)128 * dev::kSearchP16Stride, 0);
            for (unsigned c1 = 0; c1 < 128; c1++)
                for (unsigned c2 = 0; c2 < 128; c2++) p16[c1 * dev::kSearchP16Stride + c2] = (uint16_t)(4 * s2.pair_col[c1 * 128 + c2]);
            auto lay = [&](const std::vector<uint32_t> &src) {
                std::vector<uint32_t> T(src.size());
                for (size_t i = 0; i < src.size(); i++) T[i] = (src[i] & 0xffffffu) * s2.ncols * 4u | (src[i] >> 24) << 28;
                return T;
            };
            const std::vector<uint32_t> T = lay(s2.first), TA = lay(s2.all);
            oP = put(p16.data(), p16.size() * 2);
            oT = put(T.data(), T.size() * 4);
            oTA = put(TA.data(), TA.size() * 4);
        }
        {
            std::vector<uint16_t> rv(((size_t)NR * K + 1) & ~(size_t)1, 0);
            for (size_t i = 0; i < (size_t)NR * K; i++) { const uint16_t nx = search_rev.next[i]; rv[i] = (uint16_t)(nx | (search_rev.accepting[nx] ? 0x8000u : 0u)); }
            oRV = put(rv.data(), rv.size() * 2);
        }
        HIP_TRY(hipMalloc(&t.blob, host.size() + 16));
        hipError_t e = hipMemcpy(t.blob, host.data(), host.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(t.blob); return hip_fail(e, "search table upload"); }
        const uint8_t *base = static_cast<const uint8_t *>(t.blob);
        if (t.chunk.in_global) {
            t.chunk.P16 = reinterpret_cast<const uint16_t *>(base + oP);
            t.chunk.G2 = reinterpret_cast<const uint32_t *>(base + oT); t.chunk.G2_all = reinterpret_cast<const uint32_t *>(base + oTA);
        } else {
            t.chunk.P8 = base + oP;
            t.chunk.T2 = reinterpret_cast<const uint16_t *>(base + oT); t.chunk.T2_all = reinterpret_cast<const uint16_t *>(base + oTA);
        }
        t.chunk.rev = reinterpret_cast<const uint16_t *>(base + oRV);
        t.chunk.cls = base + oC;
        auto ins = search_on_device.emplace(device, t);
        *out = &ins.first->second.chunk;
        return RRX_OK;
    }

    // Upload the program for `device` once; returns the device-side descriptors.
    int tables(int device, const DeviceTables **out) const {
        std::lock_guard<std::mutex> lock(mu);
        auto it = on_device.find(device);
        if (it != on_device.end()) { *out = &it->second; return RRX_OK; }
        HIP_TRY(hipSetDevice(device));
        DeviceTables t;
        std::vector<uint8_t> host;
        auto put = [&](const void *p, size_t n) { size_t off = (host.size() + 15) & ~(size_t)15; host.resize(off + n); std::memcpy(host.data() + off, p, n); return off; };
        size_t oB = 0, oX = 0, oC = 0, oN = 0, oA = 0, oT = 0, oL = 0, oP2 = 0, oT2 = 0;
        size_t oM = 0;
        size_t oCL = 0, oCP = 0, oXI = 0, oXO = 0, oXT = 0;
        if (engine == RRX_ENGINE_NFA_BLOCK || engine == RRX_ENGINE_NFA_SPARSE) {
            // wave-resident form: 64 lanes x WL words, a B row per byte value (+ the line-mode '\n' row), exception edges as CSR
            const uint32_t W = nfa_block.W, N = nfa_block.nbits;
            // (word w of the set sits at flat index w - dense form: lane w / WL, index w % WL; sparse form: row w / 64, lane w % 64)
            const uint32_t WL = engine == RRX_ENGINE_NFA_SPARSE ? dev::sparse_rows(W) : dev::wave_words_per_lane(W);
            const uint32_t WP = 64 * WL;
            std::vector<uint32_t> M((size_t)3 * WP, 0), B((size_t)257 * WP, 0);
            const std::vector<uint32_t> *src[3] = {&nfa_block.fin, &nfa_block.self, &nfa_block.excm};
            for (int k = 0; k < 3; k++) for (uint32_t w = 0; w < W; w++) M[(size_t)k * WP + w] = (*src[k])[w];
            for (uint32_t c = 1; c < 128; c++)                            // 0x00 and >= 0x80: empty rows
                for (uint32_t w = 0; w < W; w++) B[(size_t)c * WP + w] = nfa_block.B[(size_t)c * W + w];
            B[(size_t)256 * WP] = 1u;                                     // '\n' in line mode: {position 0}
            std::vector<uint32_t> xt = nfa_block.xtgt;
            if (xt.empty()) xt.push_back(0);
            oM = put(M.data(), M.size() * 4);
            oB = put(B.data(), B.size() * 4);
            oXO = put(nfa_block.xoff.data(), nfa_block.xoff.size() * 4);
            oXT = put(xt.data(), xt.size() * 4);
            t.block.WL = WL; t.block.nbits = N;
            if (engine == RRX_ENGINE_NFA_SPARSE) {                        // rows per byte class too (LDS-resident when they fit)
                const uint32_t K = trimmed.ncls;
                std::vector<uint32_t> BC((size_t)K * WP, 0);
                for (uint32_t k = 1; k < K; k++)
                    for (uint32_t w = 0; w < W; w++) BC[(size_t)k * WP + w] = nfa_block.B[(size_t)trimmed.cls_rep[k] * W + w];
                oCL = put(BC.data(), BC.size() * 4);
                oCP = put(trimmed.cls, 256);
                t.block.ncls = K;
            }
            for (uint32_t w = 0; w < W; w++) {
                if (nfa_block.self[w]) t.block.self_words |= 1u << (w % WL);
                if (nfa_block.excm[w]) t.block.exc_words |= 1u << (w % WL);
            }
        } else if (engine == RRX_ENGINE_NFA_WAVE) {
            // group-cooperative form: G lanes x K words (device.hpp: group_geometry), a B row per byte CLASS
            const uint32_t W = nfa_wave.W, N = nfa_wave.nbits;
            uint32_t G = 0, K = 0;
            if (!dev::group_geometry(N, &G, &K)) return fail(RRX_ERR_UNSUPPORTED, "automaton too large for the group-cooperative engine");
            const uint32_t WP = G * K, NC = trimmed.ncls;
            std::vector<uint32_t> M((size_t)3 * WP, 0), B((size_t)NC * WP, 0);
            const std::vector<uint32_t> *src[3] = {&nfa_wave.fin, &nfa_wave.self, &nfa_wave.excm};
            for (int k = 0; k < 3; k++) for (uint32_t w = 0; w < W; w++) M[(size_t)k * WP + w] = (*src[k])[w];
            for (uint32_t cl = 1; cl < NC; cl++) {                        // class 0 (0x00, >= 0x80, bytes nothing moves on): empty row
                const uint32_t c = trimmed.cls_rep[cl];
                for (uint32_t w = 0; w < W; w++) B[(size_t)cl * WP + w] = nfa_wave.B[(size_t)c * W + w];
            }
            uint8_t cmap[256];
            for (int c = 0; c < 256; c++) cmap[c] = (c == 0 || c >= 128) ? 0 : trimmed.cls[c];
            // slots (word index within a lane) that carry masks at all: a slot whose B rows are all ones on every POSITION IN USE for
            // every class >= 1 needs no AND (positions beyond nbits are never set: their row bits do not matter)
            uint32_t self_slots = 0, b_slots = 0, exc_slots = 0;
            for (uint32_t w = 0; w < WP; w++) {
                const uint32_t used = w * 32 >= N ? 0u : (N - w * 32 >= 32 ? 0xffffffffu : (1u << (N - w * 32)) - 1u);
                if (w < W && nfa_wave.self[w]) self_slots |= 1u << (w % K);
                if (w < W && nfa_wave.excm[w]) exc_slots |= 1u << (w % K);
                for (uint32_t cl = 1; cl < NC; cl++)
                    if ((B[(size_t)cl * WP + w] & used) != used) b_slots |= 1u << (w % K);
            }
            std::vector<uint16_t> xidx(N, 0xffff);
            std::vector<uint32_t> X;
            uint32_t rows = 0;
            for (uint32_t b = 0; b < N; b++) {
                if (!((nfa_wave.excm[b >> 5] >> (b & 31)) & 1u)) continue;
                xidx[b] = (uint16_t)rows++;
                X.resize((size_t)rows * WP, 0);
                for (uint32_t i = nfa_wave.xoff[b]; i < nfa_wave.xoff[b + 1]; i++) {      // (the CSR lists exist at every size, dense rows only up to 4096 positions)
                    const uint32_t tv = nfa_wave.xtgt[i];
                    X[(size_t)(rows - 1) * WP + (tv >> 5)] |= 1u << (tv & 31);
                }
            }
            if (X.empty()) X.assign(WP, 0);
            oM = put(M.data(), M.size() * 4);
            oB = put(B.data(), B.size() * 4);
            oX = put(X.data(), X.size() * 4);
            oXI = put(xidx.data(), xidx.size() * 2);
            oCP = put(cmap, 256);
            t.group.G = G; t.group.K = K; t.group.nbits = N; t.group.n_exc = rows; t.group.ncls = NC;
            t.group.self_slots = self_slots; t.group.b_slots = b_slots; t.group.exc_slots = exc_slots;
            t.group.exc_mode = (rows == 1 && xidx[0] == 0) ? 2 : 0;
        } else if (engine == RRX_ENGINE_NFA) {
            const uint32_t W = nfa.W, WP = (uint32_t)instantiated_width(W);
            std::vector<uint32_t> B((size_t)256 * WP, 0), X((size_t)nfa.nbits * WP, 0);
            for (uint32_t c = 0; c < 256; c++) for (uint32_t w = 0; w < W; w++) B[(size_t)c * WP + w] = nfa.B[(size_t)c * W + w];
            for (uint32_t b = 0; b < nfa.nbits; b++) for (uint32_t w = 0; w < W; w++) X[(size_t)b * WP + w] = nfa.X[(size_t)b * W + w];
            oB = put(B.data(), B.size() * 4);
            oX = put(X.data(), X.size() * 4);
            t.nfa.W = WP; t.nfa.nbits = nfa.nbits; t.nfa.any_exc = nfa.n_exc ? 1 : 0; t.nfa.any_carry = nfa.n_carry ? 1 : 0;
            for (uint32_t w = 0; w < W; w++) if (nfa.self[w]) t.nfa.any_self = 1;
            std::memset(&t.nfa.masks, 0, sizeof t.nfa.masks);
            for (uint32_t w = 0; w < W; w++) {
                t.nfa.masks.init[w] = nfa.init[w]; t.nfa.masks.fin[w] = nfa.fin[w]; t.nfa.masks.chain[w] = nfa.chain[w];
                t.nfa.masks.self[w] = nfa.self[w]; t.nfa.masks.excm[w] = nfa.excm[w];
                t.nfa.masks.cgrp[w] = nfa.cgrp[w]; t.nfa.masks.ctgt[w] = nfa.ctgt[w];
            }
        } else {
            oC = put(dfa.cls, 256);
            oN = put(dfa.next.data(), dfa.next.size() * 2);
            oA = put(dfa.accepting.data(), dfa.accepting.size());
            t.dfa.nstates = dfa.nstates; t.dfa.ncls = dfa.ncls; t.dfa.start = dfa.start;
            // line-mode table: entry = next row byte offset (16 bits) | nl << 16 | accept << 24; the '\n' column of
            // every row goes to the start row and carries the verdict of the line that just ended.
            const uint32_t D = dfa.nstates, K = dfa.ncls;
            const bool wide = line_wide;
            uint32_t stride = wide ? dev::kWideColumns : (K + 1);
            if (!wide && !(stride & 1)) stride++;                       // odd row stride spreads rows over LDS banks
            // Wide form: R = 2^rep interleaved copies (copy k of logical dword i at dword i*R + k), lane l reads copy
            // l % R: its reads only touch LDS banks = l (mod R), so a half-wave splits into R groups that cannot
            // conflict with each other.  Row byte offsets must stay 16-bit: D * stride * 4 * R <= 65536.
            uint32_t rep = 0;
            if (wide && !line_global) while (rep < 5 && (size_t)D * stride * 4 * (2u << rep) <= 65536) rep++;
            const uint32_t R = 1u << rep;
            std::vector<uint32_t> T((size_t)D * stride, 0);
            uint8_t lcls[256];
            for (int c = 0; c < 256; c++) lcls[c] = dfa.cls[c];
            lcls['\n'] = (uint8_t)K;                                     // own column for the line terminator
            const uint32_t row_bytes = line_global ? stride : stride * 4;   // global form: entry indices, not byte offsets
            const int nl_bit = line_global ? 30 : 16, acc_bit = line_global ? 31 : 24;
            for (uint32_t d = 0; d < D; d++) {
                uint32_t *row = &T[(size_t)d * stride];
                const uint32_t nl_entry = dfa.start * row_bytes | 1u << nl_bit | (dfa.accepting[d] ? 1u << acc_bit : 0u);
                if (wide) {
                    for (uint32_t c = 0; c < 128; c++) row[c] = (uint32_t)dfa.next[(size_t)d * K + dfa.cls[c]] * row_bytes;
                    row['\n'] = nl_entry;
                    row[128] = 0;
                } else {
                    for (uint32_t k = 0; k < K; k++) row[k] = (uint32_t)dfa.next[(size_t)d * K + k] * row_bytes;
                    row[K] = nl_entry;
                }
            }
            if (R > 1) {                                                 // interleave the copies; offsets scale by R
                std::vector<uint32_t> TR(T.size() * R);
                for (size_t i = 0; i < T.size(); i++)
                    for (uint32_t k = 0; k < R; k++) TR[i * R + k] = (T[i] & 0xffffu) * R + 4 * k + (T[i] & 0xffff0000u);
                T.swap(TR);
            }
            oT = put(T.data(), T.size() * 4);
            oL = put(lcls, 256);
            if (has_dfa2) {
                std::vector<uint32_t> T2;
                std::vector<uint16_t> P;
                build_dfa2_arrays(T2, P, t.dfa2);
                oP2 = put(P.data(), P.size() * 2);
                oT2 = put(T2.data(), T2.size() * 4);
            }
            t.line.nrows = D; t.line.stride = stride * R; t.line.start_off = dfa.start * row_bytes * R; t.line.wide = wide ? 1 : 0;
            t.line.rep_log2 = rep;
            t.line.in_global = line_global ? 1 : 0;
        }
        HIP_TRY(hipMalloc(&t.blob, host.size() + 16));
        {
            const hipError_t up = hipMemcpy(t.blob, host.data(), host.size(), hipMemcpyHostToDevice);
            if (up != hipSuccess) { (void)hipFree(t.blob); return hip_fail(up, "device program upload"); }
        }
        const uint8_t *base = static_cast<const uint8_t *>(t.blob);
        if (engine == RRX_ENGINE_NFA_BLOCK || engine == RRX_ENGINE_NFA_SPARSE) {
            t.block.masks = reinterpret_cast<const uint32_t *>(base + oM);
            t.block.Bbyte = reinterpret_cast<const uint32_t *>(base + oB);
            if (engine == RRX_ENGINE_NFA_SPARSE) { t.block.Bcls = reinterpret_cast<const uint32_t *>(base + oCL); t.block.cls = base + oCP; }
            t.block.xoff = reinterpret_cast<const uint32_t *>(base + oXO);
            t.block.xtgt = reinterpret_cast<const uint32_t *>(base + oXT);
        } else if (engine == RRX_ENGINE_NFA_WAVE) {
            t.group.masks = reinterpret_cast<const uint32_t *>(base + oM);
            t.group.Bcls = reinterpret_cast<const uint32_t *>(base + oB);
            t.group.cls = base + oCP;
            t.group.X = reinterpret_cast<const uint32_t *>(base + oX);
            t.group.xidx = reinterpret_cast<const uint16_t *>(base + oXI);
        } else if (engine == RRX_ENGINE_NFA) {
            t.nfa.B = reinterpret_cast<const uint32_t *>(base + oB);
            t.nfa.X = reinterpret_cast<const uint32_t *>(base + oX);
        } else {
            t.dfa.cls = base + oC;
            t.dfa.next = reinterpret_cast<const uint16_t *>(base + oN);
            t.dfa.acc = base + oA;
            t.line.table = reinterpret_cast<const uint32_t *>(base + oT);
            t.line.cls = base + oL;
            if (has_dfa2) {
                t.dfa2.P = reinterpret_cast<const uint16_t *>(base + oP2);
                t.dfa2.T2 = reinterpret_cast<const uint32_t *>(base + oT2);
            }
        }
        auto ins = on_device.emplace(device, t);
        *out = &ins.first->second;
        return RRX_OK;
    }
};

struct rrx_corpus {
    int device = 0;
    const uint8_t *d_bytes = nullptr;
    size_t nbytes = 0, nstripes = 0, nlines = 0;
    uint32_t stripe = 0;            // bytes per lane for this corpus
    uint32_t *d_counts = nullptr;   // [nstripes] newlines per stripe, then one flags word
    uint64_t *d_base = nullptr;     // [nstripes+1] exclusive prefix
    bool has_high = false;          // some byte >= 0x80 occurs
    // A sample of the text as the batch kernel's half-waves see it - the first kSampleBytes bytes of kSampleGroups x 32
    // consecutive stripes, lane-major, in pinned host memory - taken with the index on large corpora: what a table engine
    // orders its table by at its first match (order_dfa2).  nullptr: none.
    uint8_t *h_sample = nullptr;
    uint32_t sample_lanes = 0;
    // search only: offset of the first byte of every line, built on the first search of this corpus
    mutable std::mutex mu;
    mutable uint64_t *d_line_off = nullptr;     // [nlines + 1]
    // stripe-wise search: newline prefix per search chunk (the stripe index itself when the stripe is that size)
    mutable uint64_t *d_chunk_base = nullptr;   // [nchunks + 1 + scan scratch]; owned unless it aliases d_base
    mutable size_t nchunks = 0;
    mutable void *d_all_scratch = nullptr;      // rrx_search_all: per-chunk status words, total, ticket (zeroed per call)
};

static constexpr uint32_t kSampleGroups = 8, kSampleBytes = 256;     // 8 x 32 lanes x 128 pair steps = 1024 half-waves, 64 KiB
static constexpr size_t kSampleMinCorpus = (size_t)64 << 20;         // smaller corpora: the order search (tens of ms) would not pay

static constexpr size_t kLongStringBytes = 32 * 1024;   // shorter single strings stay on one lane (NFA engines)
// Table engines: the chunk maps by convergence cost a handful of short launches (60-80 us), a sequential lane 94 ns per byte
// (tools/probe/facade_latency.py: 1.9 ms for 20 KB against 59 us; 130 us for 1 KB): from 1 KiB on the chunks win.
static constexpr size_t kLongStringBytesTable = 1024;
static constexpr uint32_t kLongNfaMaxBits = 256;        // NFA engines: chunk relations cost bytes x positions lane steps

extern "C" {

const char *rrx_last_error(void) { return g_err.c_str(); }

int rrx_compile_ex(const char *pattern, int engine, rrx_regex **out) {
    if (!pattern || !out) return fail(RRX_ERR_ARG, "null argument");
    if (engine < RRX_ENGINE_AUTO || (engine > RRX_ENGINE_DFA2 && engine != RRX_ENGINE_NFA_BLOCK && engine != RRX_ENGINE_NFA_SPARSE)) return fail(RRX_ERR_ARG, "unknown engine");
    *out = nullptr;
    rrx_regex *re = new rrx_regex();
    try {
        re->pattern = pattern;
        re->requested_engine = engine;
        re->ref = build_reference_automaton(re->pattern);
        re->trimmed = trim(re->ref);
        const Reduced red = reduce(re->trimmed);
        if (engine == RRX_ENGINE_AUTO || engine == RRX_ENGINE_NFA) re->has_nfa = lower_nfa(red, dev::kMaxNfaWords * 32, re->nfa, /*allow_carry=*/true, /*gaps=*/true);
        if (engine != RRX_ENGINE_NFA && engine != RRX_ENGINE_NFA_WAVE && engine != RRX_ENGINE_NFA_BLOCK && engine != RRX_ENGINE_NFA_SPARSE) {     // (DFA, DFA_GLOBAL, DFA2, AUTO)
            re->has_dfa = lower_dfa(red, kMaxSubsetStates, re->dfa);
            if (re->has_dfa) {
                re->line_wide = re->dfa.nstates <= dev::kWideMaxStates && engine != RRX_ENGINE_DFA_GLOBAL;
                const size_t classed_entries = (size_t)re->dfa.nstates * (re->dfa.ncls + 2);
                re->line_global = engine == RRX_ENGINE_DFA_GLOBAL || (!re->line_wide && classed_entries > dev::kClassedMaxEntries);
                if (re->line_global && classed_entries >= ((size_t)1 << 24)) re->has_dfa = false;
                // stride-2 form: when the table (rows of distinct pair columns) fits next to the 32 KiB pair table
                if (re->has_dfa && !re->line_global && engine != RRX_ENGINE_DFA && re->dfa.nstates <= 4096) {
                    re->has_dfa2 = lower_dfa2(re->dfa, 1024, re->dfa2) &&
                                   (size_t)re->dfa2.nstates * (re->dfa2.ncols | 1u) * 4 <= dev::kDfa2MaxTable;
                }
            }
        }
        // the wave-cooperative form: when asked for, or as the last resort of AUTO
        if (engine == RRX_ENGINE_NFA_WAVE || (engine == RRX_ENGINE_AUTO && !re->has_nfa && !re->has_dfa))
            re->has_wave = lower_nfa(red, dev::kGroupMaxBits, re->nfa_wave, /*allow_carry=*/false, /*gaps=*/true);
        // the wave-resident form (any automaton up to 65536 positions): when asked for, or when nothing else took it
        if (engine == RRX_ENGINE_NFA_BLOCK || engine == RRX_ENGINE_NFA_SPARSE || (engine == RRX_ENGINE_AUTO && !re->has_nfa && !re->has_dfa && !re->has_wave))
            re->has_block = lower_nfa(red, dev::kBlockMaxBits, re->nfa_block, /*allow_carry=*/false, /*gaps=*/true);
    } catch (const PatternError &e) {
        delete re;
        return fail(RRX_ERR_PATTERN, e.what());
    } catch (const BudgetError &e) {
        delete re;
        return fail(RRX_ERR_UNSUPPORTED, e.what());
    } catch (const std::exception &e) {
        delete re;
        return fail(RRX_ERR_PATTERN, std::string("internal: ") + e.what());
    }
    // AUTO: the LDS-resident table when it fits, else the register-resident NFA, else the table in global memory
    if (engine == RRX_ENGINE_NFA) re->engine = re->has_nfa ? RRX_ENGINE_NFA : 0;
    else if (engine == RRX_ENGINE_DFA || engine == RRX_ENGINE_DFA_GLOBAL) re->engine = re->has_dfa ? RRX_ENGINE_DFA : 0;
    else if (engine == RRX_ENGINE_NFA_WAVE) re->engine = re->has_wave ? RRX_ENGINE_NFA_WAVE : 0;
    else if (engine == RRX_ENGINE_NFA_BLOCK || engine == RRX_ENGINE_NFA_SPARSE) re->engine = re->has_block ? engine : 0;
    else if (engine == RRX_ENGINE_DFA2) re->engine = re->has_dfa2 ? RRX_ENGINE_DFA : 0;
    else re->engine = (re->has_dfa && !re->line_global) ? RRX_ENGINE_DFA : re->has_nfa ? RRX_ENGINE_NFA : re->has_dfa ? RRX_ENGINE_DFA
                      : re->has_wave ? RRX_ENGINE_NFA_WAVE : re->has_block ? RRX_ENGINE_NFA_BLOCK : 0;
    if (!re->engine) {
        char msg[200];
        std::snprintf(msg, sizeof msg, "automaton too large for the requested engine (%u useful states)", re->trimmed.n);
        delete re;
        return fail(RRX_ERR_UNSUPPORTED, msg);
    }
    *out = re;
    return RRX_OK;
}
int rrx_compile(const char *pattern, rrx_regex **out) { return rrx_compile_ex(pattern, RRX_ENGINE_AUTO, out); }
void rrx_free(rrx_regex *re) { delete re; }

uint32_t rrx_num_states(const rrx_regex *re) { return re->ref.states_n; }
int rrx_set_class(const rrx_regex *re) { return re->ref.set_class(); }
uint32_t rrx_ref_initial(const rrx_regex *re) { return re->ref.initial; }
int rrx_ref_is_final(const rrx_regex *re, uint32_t s) { return s < re->ref.states_n && re->ref.is_final[s]; }
uint32_t rrx_ref_row(const rrx_regex *re, uint32_t state, unsigned c, uint32_t *out, uint32_t cap) {
    std::vector<uint32_t> r = re->ref.row(state, c);
    for (size_t i = 0; i < r.size() && i < cap; i++) out[i] = r[i];
    return (uint32_t)r.size();
}
int rrx_engine(const rrx_regex *re) { return re->engine; }
const char *rrx_engine_name(const rrx_regex *re) {
    if (re->engine == RRX_ENGINE_NFA_WAVE) return "nfa-group-cooperative";
    if (re->engine == RRX_ENGINE_NFA_BLOCK) return "nfa-wave-resident";
    if (re->engine == RRX_ENGINE_NFA_SPARSE) return "nfa-wave-sparse";
    if (re->engine != RRX_ENGINE_DFA) return "nfa-shift-and";
    if (re->has_dfa2) return "dfa-stride2-table";      // (the byte-stride table still serves corpora with bytes >= 0x80)
    return re->line_global ? "dfa-global-table" : re->line_wide ? "dfa-wide-table" : "dfa-classed-table";
}
uint32_t rrx_useful_states(const rrx_regex *re) { return re->trimmed.n; }
int rrx_order_table(rrx_regex *re, const void *sample, uint32_t lanes, uint32_t bytes_per_lane) {
    if (!re || !sample || lanes < 32 || bytes_per_lane < 2) return fail(RRX_ERR_ARG, "sample: at least 32 lanes of 2 bytes");
    if (!re->decide_t2_order(static_cast<const uint8_t *>(sample), lanes, bytes_per_lane, /*now=*/true))
        return fail(RRX_ERR_ARG, "the table order has been decided already");
    return RRX_OK;
}
int rrx_table_order(const rrx_regex *re, double *conflict_before, double *conflict_after) {
    const TableOrderSearch::State st = re->t2_order.state();          // (one atomic read; the thread object is its owner's)
    std::lock_guard<std::mutex> lock(re->mu);
    const bool profiled = st == TableOrderSearch::kDone && re->t2_row_slot.size() == re->dfa2.nstates && re->has_dfa2 && re->t2_order_stats.half_waves;
    if (conflict_before) *conflict_before = profiled ? re->t2_order_stats.before : 0.0;
    if (conflict_after) *conflict_after = profiled ? re->t2_order_stats.after : 0.0;
    return profiled ? 1 : st == TableOrderSearch::kRunning ? 2 : 0;   // 2: the search is running
}
int rrx_learn_table(rrx_regex *re, const void *text, size_t nbytes) {
    if (!re || !text || nbytes < 2 || nbytes > ((size_t)1 << 30)) return fail(RRX_ERR_ARG, "a text sample of 2 bytes to 1 GiB");
    if (!re->sampled_eligible()) return fail(RRX_ERR_UNSUPPORTED, "a sampled table serves automata that AUTO leaves on the NFA lane engine");
    const uint8_t *p = static_cast<const uint8_t *>(text);
    bool built = false;
    if (!re->sampled_build.start([&]() { built = re->build_sampled(p, 1, (uint32_t)nbytes); }, /*background=*/false))
        return fail(RRX_ERR_ARG, "the sampled table has been decided already");
    return built ? RRX_OK : fail(RRX_ERR_UNSUPPORTED, "no sampled table for this automaton and text: none fits the device, or more than 2 % of the text's own lines leave it");
}
int rrx_sampled_table(const rrx_regex *re, uint32_t *table_states, uint32_t *open_transitions) {
    const OnceTask::State st = re->sampled_build.state();
    const bool ready = re->sampled_ready.load(std::memory_order_acquire);
    std::lock_guard<std::mutex> lock(re->mu);
    if (table_states) *table_states = ready ? re->sampled_dfa.nstates : 0;
    if (open_transitions) *open_transitions = ready ? re->sampled_stats.open_transitions : 0;
    return ready ? (re->sampled_retired.load() ? 3 : 1) : st == OnceTask::kRunning ? 2 : 0;
}
int rrx_sampled_escapes(const rrx_regex *re, int device, uint64_t *lines) {
    if (!re || !lines) return fail(RRX_ERR_ARG, "null argument");
    *lines = 0;
    std::lock_guard<std::mutex> lock(re->onepass_mu);
    auto it = re->sampled_counter.find(device);
    if (it == re->sampled_counter.end() || !it->second) return RRX_OK;            // no sampled-table launch on this device yet
    HIP_TRY(hipSetDevice(device));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpy(&v, it->second, sizeof v, hipMemcpyDeviceToHost));            // (synchronous: behind everything queued on the device)
    *lines = v;
    return RRX_OK;
}
int rrx_set_option(rrx_regex *re, int option, int64_t value) {
    if (!re) return fail(RRX_ERR_ARG, "null argument");
    if (option == RRX_OPT_BACKGROUND_ORDER) { re->opt_background_order.store(value ? 1 : 0); return RRX_OK; }
    if (option == RRX_OPT_FLUSH_SLOTS) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16 && value != 32) return fail(RRX_ERR_ARG, "flush period: 0 (automatic) or 1, 2, 4, 8, 16, 32 slots");
        re->opt_flush_slots.store((int)value);
        return RRX_OK;
    }
    if (option == RRX_OPT_SAMPLED_TABLE) { re->opt_sampled_table.store(value ? 1 : 0); return RRX_OK; }
    if (option == RRX_OPT_ITEMS_STRIDE2) { re->items_stride2.store(value ? 1 : 0); return RRX_OK; }
    if (option == RRX_OPT_SEARCH_ANCHORED) {
        std::lock_guard<std::mutex> lock(re->mu);
        if (re->search_state != 0) return fail(RRX_ERR_ARG, "the search tables of this regex are built already");
        re->opt_search_anchored.store(value ? 1 : 0);
        return RRX_OK;
    }
    if (option == RRX_OPT_UNITS_PER_WORKGROUP) {
        if (value < 0 || value > 65536) return fail(RRX_ERR_ARG, "units per workgroup: 0 (off) or 16 ... 65536");
        re->opt_units_per_wg.store(value && value < 16 ? 16 : (int)value);
        return RRX_OK;
    }
    return fail(RRX_ERR_ARG, "unknown option");
}
uint32_t rrx_byte_classes(const rrx_regex *re) { return re->trimmed.ncls; }
uint32_t rrx_words_per_set(const rrx_regex *re) { return re->has_nfa ? re->nfa.W : re->has_wave ? re->nfa_wave.W : re->has_block ? re->nfa_block.W : 0; }
int rrx_accepts_empty(const rrx_regex *re) {
    return re->has_nfa ? re->nfa.accepts_empty : re->has_wave ? re->nfa_wave.accepts_empty : re->has_block ? re->nfa_block.accepts_empty : re->dfa.accepts_empty;
}

size_t rrx_program_words(const rrx_regex *re, int kind, uint32_t *out, size_t cap) {
    std::vector<uint32_t> w;
    if ((kind == RRX_ENGINE_NFA && re->has_nfa) || (kind == RRX_ENGINE_NFA_WAVE && re->has_wave)) {
        const NfaProgram &p = kind == RRX_ENGINE_NFA ? re->nfa : re->nfa_wave;
        w = {p.W, p.nbits, p.n_exc, p.accepts_empty ? 1u : 0u};
        for (auto *v : {&p.init, &p.fin, &p.chain, &p.self, &p.excm, &p.cgrp, &p.ctgt, &p.B, &p.X}) w.insert(w.end(), v->begin(), v->end());
    } else if ((kind == RRX_ENGINE_NFA_BLOCK || kind == RRX_ENGINE_NFA_SPARSE) && re->has_block) {
        const NfaProgram &p = re->nfa_block;
        w = {p.W, p.nbits, p.n_exc, p.accepts_empty ? 1u : 0u};
        for (auto *v : {&p.init, &p.fin, &p.chain, &p.self, &p.excm, &p.cgrp, &p.ctgt, &p.B, &p.xoff, &p.xtgt}) w.insert(w.end(), v->begin(), v->end());
    } else if (kind == RRX_PROGRAM_SEARCH_LINE) {
        std::lock_guard<std::mutex> lock(re->mu);
        if (re->build_search() || !re->search_line.nrows) return 0;
        const SearchLineProgram &d = re->search_line;
        w = {d.nrows, d.ncols, d.start, d.skip};
        for (int c = 0; c < 256; c++) w.push_back(c == '\n' ? d.ncols - 1 : re->search_fwd.cls[c]);
        w.insert(w.end(), d.table.begin(), d.table.end());
    } else if (kind == RRX_PROGRAM_SEARCH_LINE2) {
        std::lock_guard<std::mutex> lock(re->mu);
        if (re->build_search() || !re->search_line2.nrows) return 0;
        const SearchLine2Program &d = re->search_line2;
        w = {d.nrows, d.ncols, d.start, d.skip, re->chunk_proto.nrows ? (re->chunk_proto.in_global ? 2u : 1u) : 0u};
        for (uint16_t c : d.pair_col) w.push_back(c);
        w.insert(w.end(), d.first.begin(), d.first.end());
        w.insert(w.end(), d.all.begin(), d.all.end());
    } else if (kind == RRX_PROGRAM_DFA2_ORDER && re->has_dfa2) {
        std::lock_guard<std::mutex> lock(re->mu);
        if (re->t2_row_slot.size() != re->dfa2.nstates || re->t2_col_slot.size() != re->dfa2.ncols) return 0;
        w = {re->dfa2.nstates, re->dfa2.ncols};
        w.insert(w.end(), re->t2_row_slot.begin(), re->t2_row_slot.end());
        w.insert(w.end(), re->t2_col_slot.begin(), re->t2_col_slot.end());
    } else if (kind == RRX_PROGRAM_SAMPLED_DFA || kind == RRX_PROGRAM_SAMPLED_DFA2) {
        if (!re->sampled_ready.load(std::memory_order_acquire)) return 0;
        std::lock_guard<std::mutex> lock(re->mu);
        if (kind == RRX_PROGRAM_SAMPLED_DFA) {              // the DFA layout, then the escaped flag per state
            const DfaProgram &d = re->sampled_dfa;
            w = {d.nstates, d.ncls, d.start, d.accepts_empty ? 1u : 0u};
            for (int c = 0; c < 256; c++) w.push_back(d.cls[c]);
            for (uint8_t a : d.accepting) w.push_back(a);
            for (uint16_t n : d.next) w.push_back(n);
            for (uint8_t e : d.escaped) w.push_back(e);
        } else {                                            // the stride-2 layout (entries: next | result bits << 16 | verdict pairs << 24)
            const Dfa2Program &d = re->sampled_dfa2;
            w = {d.nstates, d.ncols, d.start, d.accepts_empty ? 1u : 0u};
            for (uint16_t c : d.pair_col) w.push_back(c);
            w.insert(w.end(), d.next2.begin(), d.next2.end());
        }
    } else if (kind == RRX_PROGRAM_DFA2_ITEMS && re->has_dfa2) {
        if (re->items2_program()) {
            const Dfa2Program &d = re->items2_prog;
            w = {d.nstates, d.ncols, d.start, d.accepts_empty ? 1u : 0u, d.pair_dim};
            for (uint16_t c : d.pair_col) w.push_back(c);
            w.insert(w.end(), d.next2.begin(), d.next2.end());
        }
    } else if (kind == RRX_ENGINE_DFA2 && re->has_dfa2) {
        const Dfa2Program &d = re->dfa2;
        w = {d.nstates, d.ncols, d.start, d.accepts_empty ? 1u : 0u};
        for (uint16_t c : d.pair_col) w.push_back(c);
        w.insert(w.end(), d.next2.begin(), d.next2.end());
    } else if ((kind == RRX_ENGINE_DFA && re->has_dfa) || kind == RRX_PROGRAM_SEARCH_FWD || kind == RRX_PROGRAM_SEARCH_REV) {
        if (kind != RRX_ENGINE_DFA) {
            std::lock_guard<std::mutex> lock(re->mu);
            if (re->build_search()) return 0;
        }
        const DfaProgram &d = kind == RRX_ENGINE_DFA ? re->dfa : kind == RRX_PROGRAM_SEARCH_FWD ? re->search_fwd : re->search_rev;
        w = {d.nstates, d.ncls, d.start, d.accepts_empty ? 1u : 0u};
        for (int c = 0; c < 256; c++) w.push_back(d.cls[c]);
        for (uint8_t a : d.accepting) w.push_back(a);
        for (uint16_t n : d.next) w.push_back(n);
    }
    for (size_t i = 0; i < w.size() && i < cap; i++) out[i] = w[i];
    return w.size();
}

int rrx_corpus_create(int device, const void *d_bytes, size_t nbytes, void *stream, rrx_corpus **out) {
    return rrx_corpus_create_ex(device, d_bytes, nbytes, 0, stream, out);
}

int rrx_corpus_create_ex(int device, const void *d_bytes, size_t nbytes, uint32_t stripe_bytes, void *stream, rrx_corpus **out) {
    if (!out || (nbytes && !d_bytes)) return fail(RRX_ERR_ARG, "null argument");
    if (stripe_bytes && (stripe_bytes < dev::kMinStripe || stripe_bytes > dev::kMaxStripe || (stripe_bytes & (stripe_bytes - 1))))
        return fail(RRX_ERR_ARG, "stripe must be a power of two in [1024, 16384]");
    if (reinterpret_cast<uintptr_t>(d_bytes) & 15) return fail(RRX_ERR_ARG, "corpus base must be 16-byte aligned");
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    rrx_corpus *c = new rrx_corpus();
    c->device = device;
    c->d_bytes = static_cast<const uint8_t *>(d_bytes);
    c->nbytes = nbytes;
    c->stripe = stripe_bytes ? stripe_bytes : dev::pick_stripe(nbytes);
    // automatic choice on a large corpus: the line length is taken from its first 4 MiB first (two small launches), so
    // that a corpus of very short or very long lines is indexed once, not twice (the check below still stands)
    constexpr size_t kSample = (size_t)4 << 20;
    if (!stripe_bytes && nbytes >= 16 * kSample) {
        rrx_corpus *sample = nullptr;
        if (rrx_corpus_create_ex(device, d_bytes, kSample, dev::pick_stripe(kSample), stream, &sample) == RRX_OK && sample) {
            if (sample->nlines) c->stripe = dev::stripe_for_lines(nbytes, kSample / sample->nlines);
            rrx_corpus_free(sample);
        }
    }
    c->nstripes = (nbytes + c->stripe - 1) / c->stripe;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&c->d_counts), (c->nstripes + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->d_base), (c->nstripes + 1 + dev::scan_scratch_words(c->nstripes)) * sizeof(uint64_t));
    if (e != hipSuccess) { rrx_corpus_free(c); return hip_fail(e, "hipMalloc(line index)"); }
    uint32_t *d_flags = c->d_counts + c->nstripes;
    e = hipMemsetAsync(d_flags, 0, sizeof(uint32_t), (hipStream_t)stream);
    if (e != hipSuccess) { rrx_corpus_free(c); return hip_fail(e, "hipMemsetAsync(flags)"); }
    int rc = dev::count_newlines_per_stripe(c->d_bytes, nbytes, c->stripe, c->d_counts, c->nstripes, d_flags, stream);
    if (!rc) rc = dev::scan_counts(c->d_counts, c->d_base, c->d_base + c->nstripes + 1, c->nstripes, stream);
    if (rc) { rrx_corpus_free(c); return hip_fail((hipError_t)rc, "line index launch"); }
    MailboxGuard mail;
    if (int mrc = mailbox_acquire(device, &mail.m)) { rrx_corpus_free(c); return mrc; }
    if (nbytes >= kSampleMinCorpus && c->nstripes >= 64 * kSampleGroups &&
        hipHostMalloc(reinterpret_cast<void **>(&c->h_sample), (size_t)kSampleGroups * 32 * kSampleBytes, hipHostMallocDefault) == hipSuccess) {
        c->sample_lanes = kSampleGroups * 32;
        for (uint32_t g = 0; g < kSampleGroups; g++) {            // group g: 32 consecutive stripes, the groups spread over the corpus
            const size_t first_stripe = (size_t)g * (c->nstripes / kSampleGroups);
            if (hipMemcpy2DAsync(c->h_sample + (size_t)g * 32 * kSampleBytes, kSampleBytes, c->d_bytes + first_stripe * c->stripe, c->stripe,
                                 kSampleBytes, 32, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipStreamSynchronize((hipStream_t)stream);      // the copies of the groups before this one may still be writing the buffer
                (void)hipHostFree(c->h_sample); c->h_sample = nullptr; c->sample_lanes = 0;
                break;
            }
        }
    }
    mail.stream = (hipStream_t)stream; mail.queued = true;
    rc = dev::mail_results(c->d_base + c->nstripes, d_flags, nbytes ? c->d_bytes + nbytes - 1 : nullptr, mail.m.dev, stream);
    if (rc) { rrx_corpus_free(c); return hip_fail((hipError_t)rc, "line index launch"); }
    e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) { rrx_corpus_free(c); return hip_fail(e, "line index readback"); }
    mail.drained = true;
    const uint64_t total = mail.m.host[0];
    const uint32_t flags = (uint32_t)mail.m.host[1];
    const uint8_t last = nbytes ? (uint8_t)mail.m.host[2] : (uint8_t)'\n';
    c->has_high = (flags & 1u) != 0;
    c->nlines = (size_t)total + ((nbytes && last != '\n') ? 1 : 0);
    // with the line count known: the stripe this corpus wants (stripe_for_lines); if it is another one, index once more
    if (!stripe_bytes && c->nlines) {
        const uint32_t want = dev::stripe_for_lines(nbytes, nbytes / c->nlines);
        if (want != c->stripe) {
            rrx_corpus_free(c);
            return rrx_corpus_create_ex(device, d_bytes, nbytes, want, stream, out);
        }
    }
    *out = c;
    return RRX_OK;
}
size_t rrx_corpus_num_lines(const rrx_corpus *c) { return c->nlines; }
size_t rrx_corpus_num_bytes(const rrx_corpus *c) { return c->nbytes; }
uint32_t rrx_corpus_stripe_bytes(const rrx_corpus *c) { return c->stripe; }
void rrx_corpus_free(rrx_corpus *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->d_base) (void)hipFree(c->d_base);
    if (c->d_line_off) (void)hipFree(c->d_line_off);
    if (c->d_chunk_base && c->d_chunk_base != c->d_base) (void)hipFree(c->d_chunk_base);
    if (c->d_all_scratch) (void)hipFree(c->d_all_scratch);
    if (c->h_sample) (void)hipHostFree(c->h_sample);
    delete c;
}

size_t rrx_corpus_bitmap_words(const rrx_corpus *c) { return (c->nlines + 31) / 32; }

// The batch entry on the sampled table: the stride-2 kernel with two result bits per line (accepted, escaped), the two bitmaps
// taken apart, the escaped lines decided by the NFA lane engine.  Scratch (the wide bitmap, the escaped bitmap, a counter) is
// the regex' event-ordered per-device buffer; everything is queued on `stream`, nothing is read back.
static int match_corpus_sampled(const rrx_regex *re, const rrx_corpus *c, const DeviceTables *t, uint32_t *d_accept_bits, void *stream) {
    dev::Dfa2Device d2;
    int rc = re->sampled_tables(c->device, &d2);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t words = rrx_corpus_bitmap_words(c);
    const size_t wide_bytes = (2 * words * sizeof(uint32_t) + 15) & ~(size_t)15, esc_bytes = (words * sizeof(uint32_t) + 15) & ~(size_t)15;
    const size_t cap = std::max<size_t>(words / 2, 1024);                        // listed escaped lines: 1.5 % of the lines (then: the walk over the stripes)
    std::lock_guard<std::mutex> lock(re->onepass_mu);
    void *buf = nullptr;
    rc = re->onepass_for(c->device, wide_bytes + esc_bytes + 16 + cap * sizeof(uint64_t), &buf, st);
    if (rc) return rc;
    uint32_t *wide = static_cast<uint32_t *>(buf);
    uint32_t *escaped = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(buf) + wide_bytes);
    unsigned long long *total = reinterpret_cast<unsigned long long *>(static_cast<uint8_t *>(buf) + wide_bytes + esc_bytes);
    uint64_t *list = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(buf) + wide_bytes + esc_bytes + 16);
    re->sampled_counter[c->device] = total;
    constexpr uint32_t kSlots = rrx_regex::kSampledRelearns + 1;
    if (!re->h_sampled_seen && hipHostMalloc(reinterpret_cast<void **>(&re->h_sampled_seen), kSlots * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess)
        for (uint32_t k = 0; k < kSlots; k++) re->h_sampled_seen[k] = 0;
    unsigned long long *const seen = re->h_sampled_seen ? re->h_sampled_seen + (re->sampled_gen < kSlots ? re->sampled_gen : kSlots - 1) : nullptr;
    if (seen) {
        // what the last FINISHED launch counted, against the size of the last launch queued (the same corpus in a scan loop; otherwise a hint)
        const unsigned long long esc_seen = seen[0];
        if (re->sampled_prev_lines >= 1024 && esc_seen * 20 > re->sampled_prev_lines) re->sampled_retired.store(true);   // > 5 % of the lines: the wrong table for this text
    }
    hipError_t he = hipMemsetAsync(wide, 0, wide_bytes, st);                     // (the kernel merges words with atomic OR)
    if (he == hipSuccess) he = hipMemsetAsync(total, 0, 16, st);
    int e = he != hipSuccess ? (int)he : dev::match_stripes_dfa2_two_bit(d2, c->d_bytes, c->nbytes, c->stripe, c->d_base, c->nstripes, wide, stream);
    if (!e) e = dev::split_two_bit(wide, c->nlines, d_accept_bits, escaped, total, list, cap, stream);
    if (!e) e = dev::recheck_escaped_nfa(t->nfa, c->d_bytes, c->nbytes, c->stripe, c->d_base, c->nstripes, escaped, c->nlines, list, total, cap, d_accept_bits, stream);
    if (!e && seen) {                                                            // behind the kernels: the count into pinned memory (nobody waits for it)
        if (hipMemcpyAsync(seen, total, sizeof(unsigned long long), hipMemcpyDeviceToHost, st) != hipSuccess) (void)hipGetLastError();
        re->sampled_prev_lines = c->nlines;
    }
    const int rc2 = re->onepass_done(c->device, st);
    if (e) return hip_fail((hipError_t)e, "sampled-table launch");
    return rc2;
}

int rrx_match_corpus(const rrx_regex *re, const rrx_corpus *c, uint32_t *d_accept_bits, void *stream) {
    if (!re || !c || (c->nlines && !d_accept_bits)) return fail(RRX_ERR_ARG, "null argument");
    if (!re->t2_order.decided() && c->h_sample && !c->has_high) (void)re->decide_t2_order(c->h_sample, c->sample_lanes, kSampleBytes, /*now=*/false);
    const DeviceTables *t;
    int rc = re->tables(c->device, &t);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->nlines) return RRX_OK;
    // the sampled table: its build starts at the first match against a corpus that carries a text sample (in the background;
    // this launch and the next ones run on the NFA engine until it is in), and serves corpora without bytes >= 0x80
    if (re->sampled_eligible() && re->opt_sampled_table.load() && !c->has_high) {
        if (!re->sampled_build.decided() && c->h_sample) {
            auto text = std::make_shared<std::vector<uint8_t>>(c->h_sample, c->h_sample + (size_t)c->sample_lanes * kSampleBytes);
            const uint32_t pieces = c->sample_lanes;
            (void)re->sampled_build.start([re, text, pieces]() { (void)re->build_sampled(text->data(), pieces, kSampleBytes); },
                                          /*background=*/re->opt_background_order.load() != 0);
        }
        if (re->sampled_ready.load(std::memory_order_acquire) && re->sampled_retired.load() && c->h_sample)
            re->relearn_sampled(c->h_sample, c->sample_lanes, kSampleBytes);     // (this launch and the next ones: the NFA engine, until the new table is in)
        if (re->sampled_ready.load(std::memory_order_acquire) && !re->sampled_retired.load()) return match_corpus_sampled(re, c, t, d_accept_bits, stream);
    }
    // the kernel merges words with atomic OR: start from an all-zero bitmap
    HIP_TRY(hipMemsetAsync(d_accept_bits, 0, rrx_corpus_bitmap_words(c) * sizeof(uint32_t), (hipStream_t)stream));
    int e = re->engine == RRX_ENGINE_NFA_SPARSE
                ? dev::match_stripes_sparse_nfa(t->block, c->d_bytes, c->nbytes, c->stripe, c->d_base, c->nstripes, d_accept_bits, stream)
            : re->engine == RRX_ENGINE_NFA_BLOCK
                ? dev::match_stripes_wave_nfa(t->block, c->d_bytes, c->nbytes, c->stripe, c->d_base, c->nstripes, d_accept_bits, stream)
            : re->engine == RRX_ENGINE_NFA_WAVE
                ? dev::match_stripes_group_nfa(t->group, c->d_bytes, c->nbytes, c->stripe, c->d_base, c->nstripes, d_accept_bits, stream)
            : re->engine == RRX_ENGINE_NFA
                ? dev::match_stripes_nfa(t->nfa, c->d_bytes, c->nbytes, c->stripe, c->d_base, c->nstripes, d_accept_bits, stream)
            : (re->has_dfa2 && !c->has_high)
                ? (re->opt_units_per_wg.load()
                       ? dev::match_units_dfa2(re->dfa2_device(t), c->d_bytes, c->nbytes, c->stripe, c->d_base, c->nstripes, d_accept_bits, (uint32_t)re->opt_units_per_wg.load(), stream)
                       : dev::match_stripes_dfa2(re->dfa2_device(t), c->d_bytes, c->nbytes, c->stripe, c->d_base, c->nstripes, d_accept_bits, stream,
                                                 re->opt_flush_slots.load() ? (uint32_t)re->opt_flush_slots.load() - 1u : dev::flush_mask_for(c->nbytes, c->nlines)))
                : dev::match_stripes_dfa(t->line, c->has_high, c->d_bytes, c->nbytes, c->stripe, c->d_base, c->nstripes, d_accept_bits, stream);
    if (e) return hip_fail((hipError_t)e, "match_stripes launch");
    return RRX_OK;
}

// One-shot entry: a device-resident buffer that nobody has indexed.  With the lane engines (tables and NFA) the text is
// read ONCE: the match kernel counts the '\n' of every stripe on the side and leaves every lane's verdicts as a stream of
// its own; a scan of the counts and a small compaction kernel then put the streams at their line numbers.  The
// cooperative engines build the index first (two passes).  Synchronous: *nlines is read back.
int rrx_match_device(const rrx_regex *re, int device, const void *d_bytes, size_t nbytes, uint32_t *d_accept_bits, size_t cap_words,
                     size_t *nlines, void *stream) {
    if (!re || (nbytes && !d_bytes) || !nlines || (cap_words && !d_accept_bits)) return fail(RRX_ERR_ARG, "null argument");
    if (reinterpret_cast<uintptr_t>(d_bytes) & 15) return fail(RRX_ERR_ARG, "corpus base must be 16-byte aligned");
    *nlines = 0;
    HIP_TRY(hipSetDevice(device));
    if (!nbytes) return RRX_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // cooperative engines: two passes (index, then match) - and so does a regex that runs on its sampled table: the index pass and the
    // table kernel (1.4 + 2.1 ms per 8 GiB of URL text) are a fifth of the NFA lane engine's one pass (16.7 ms)
    const bool sampled = re->sampled_eligible() && re->opt_sampled_table.load() && re->sampled_ready.load(std::memory_order_acquire) && !re->sampled_retired.load();
    if ((re->engine != RRX_ENGINE_DFA && re->engine != RRX_ENGINE_NFA) || sampled) {
        rrx_corpus *c = nullptr;
        int rc = rrx_corpus_create(device, d_bytes, nbytes, stream, &c);
        if (rc) return rc;
        *nlines = c->nlines;
        if (rrx_corpus_bitmap_words(c) > cap_words) rc = fail(RRX_ERR_ARG, "accept bitmap too small for the number of strings");
        if (!rc) rc = rrx_match_corpus(re, c, d_accept_bits, stream);
        const hipError_t e = hipStreamSynchronize(st);                      // the index arrays of `c` are freed next
        if (!rc && e != hipSuccess) rc = hip_fail(e, "match_device");
        rrx_corpus_free(c);
        return rc;
    }
    const DeviceTables *t;
    int rc = re->tables(device, &t);
    if (rc) return rc;
    const uint8_t *bytes = static_cast<const uint8_t *>(d_bytes);
    const uint32_t stripe = dev::pick_stripe(nbytes);
    const size_t nstripes = (nbytes + stripe - 1) / stripe;
    // scratch: [counts u32 (nstripes) | flag u32 | pad] [base u64 (nstripes + 1) + scan scratch] [slabs u32]
    const size_t counts_bytes = ((nstripes + 2) * sizeof(uint32_t) + 15) & ~(size_t)15;
    const size_t base_bytes = (nstripes + 1 + dev::scan_scratch_words(nstripes)) * sizeof(uint64_t);
    const size_t slab_bytes = dev::onepass_slab_words(nstripes, stripe) * sizeof(uint32_t);
    std::lock_guard<std::mutex> lock(re->onepass_mu);
    void *buf = nullptr;
    rc = re->onepass_for(device, counts_bytes + base_bytes + slab_bytes, &buf, st);
    if (rc) return rc;
    uint32_t *d_counts = static_cast<uint32_t *>(buf);
    uint64_t *d_base = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(buf) + counts_bytes);
    uint32_t *d_slabs = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(buf) + counts_bytes + base_bytes);
    MailboxGuard mail;
    rc = mailbox_acquire(device, &mail.m);
    if (rc) return rc;
    if (cap_words) HIP_TRY(hipMemsetAsync(d_accept_bits, 0, cap_words * sizeof(uint32_t), st));
    int e = re->engine == RRX_ENGINE_NFA ? dev::match_onepass_nfa(t->nfa, bytes, nbytes, stripe, nstripes, d_counts, d_slabs, stream)
            : re->has_dfa2               ? dev::match_onepass_dfa2(re->dfa2_device(t), bytes, nbytes, stripe, nstripes, d_counts, d_slabs, stream)
                                         : dev::match_onepass_dfa(t->line, bytes, nbytes, stripe, nstripes, d_counts, d_slabs, stream);
    if (!e) e = dev::scan_counts(d_counts, d_base, d_base + nstripes + 1, nstripes, stream);
    // (words beyond the caller's bitmap are dropped by the compaction; whether there were any follows from the line count)
    if (!e) e = dev::compact_streams(d_counts, d_base, nstripes, stripe, d_slabs, d_accept_bits, cap_words, stream);
    mail.stream = st; mail.queued = true;
    if (!e) e = dev::mail_results(d_base + nstripes, nullptr, bytes + nbytes - 1, mail.m.dev, stream);
    if (e) return hip_fail((hipError_t)e, "one-pass launch");
    const hipError_t he = hipStreamSynchronize(st);
    if (he != hipSuccess) return hip_fail(he, "one-pass readback");
    mail.drained = true;
    *nlines = (size_t)mail.m.host[0] + ((uint8_t)mail.m.host[2] != '\n' ? 1 : 0);
    if ((*nlines + 31) / 32 > cap_words) return fail(RRX_ERR_ARG, "accept bitmap too small for the number of strings");
    return RRX_OK;
}

// per-line offsets of the corpus, built on the first search (cached in the corpus)
static int line_offsets(const rrx_corpus *c, void *stream) {
    {
        std::lock_guard<std::mutex> lock(c->mu);
        if (!c->d_line_off) {
            uint64_t *off = nullptr;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&off), (c->nlines + 1) * sizeof(uint64_t)));
            // entry nlines: one past the last '\n' - the kernel writes it when the corpus ends in '\n'; otherwise the
            // last line ends at the end of the data, as if a '\n' followed it
            const uint64_t past = (uint64_t)c->nbytes + 1;
            hipError_t he = hipMemcpyAsync(off + c->nlines, &past, sizeof past, hipMemcpyHostToDevice, (hipStream_t)stream);
            if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream);          // `past` leaves scope
            int le = he == hipSuccess ? dev::build_line_offsets(c->d_bytes, c->nbytes, c->stripe, c->d_base, c->nstripes, off, stream) : 0;
            if (he == hipSuccess && !le) he = hipStreamSynchronize((hipStream_t)stream);   // once per corpus: later searches may use other streams
            if (he != hipSuccess || le) { (void)hipFree(off); return he != hipSuccess ? hip_fail(he, "line offsets") : hip_fail((hipError_t)le, "line_offsets launch"); }
            c->d_line_off = off;
        }
    }
    return RRX_OK;
}

// newline prefix per search chunk (the granularity of the stripe-wise search kernel), built on the first search
static int chunk_index(const rrx_corpus *c, void *stream) {
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->d_chunk_base) return RRX_OK;
    const size_t chunk = dev::search_chunk_bytes();
    c->nchunks = (c->nbytes + chunk - 1) / chunk;
    if (c->stripe == chunk) { c->d_chunk_base = c->d_base; return RRX_OK; }
    uint32_t *counts = nullptr;
    uint64_t *base = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&counts), (c->nchunks + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&base), (c->nchunks + 1 + dev::scan_scratch_words(c->nchunks)) * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemsetAsync(counts + c->nchunks, 0, sizeof(uint32_t), (hipStream_t)stream);
    int le = 0;
    if (e == hipSuccess) le = dev::count_newlines_per_stripe(c->d_bytes, c->nbytes, (uint32_t)chunk, counts, c->nchunks, counts + c->nchunks, stream);
    if (e == hipSuccess && !le) le = dev::scan_counts(counts, base, base + c->nchunks + 1, c->nchunks, stream);
    if (e == hipSuccess && !le) e = hipStreamSynchronize((hipStream_t)stream);     // once per corpus: later searches may use other streams
    if (counts) (void)hipFree(counts);
    if (e != hipSuccess || le) { if (base) (void)hipFree(base); return e != hipSuccess ? hip_fail(e, "search chunk index") : hip_fail((hipError_t)le, "search chunk index launch"); }
    c->d_chunk_base = base;
    return RRX_OK;
}

int rrx_search_corpus(const rrx_regex *re, const rrx_corpus *c, uint32_t *d_start, uint32_t *d_end, void *stream) {
    if (!re || !c || (c->nlines && (!d_start || !d_end))) return fail(RRX_ERR_ARG, "null argument");
    const dev::SearchChunkDevice *ct;
    int rc = re->search_tables(c->device, &ct);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->nlines) return RRX_OK;
    if (!ct) {
        // the pattern accepts the empty string: the accepted substring with the smallest end is [0, 0) in every string - no
        // table, no line offsets, two fills
        HIP_TRY(hipMemsetAsync(d_start, 0, c->nlines * sizeof(uint32_t), (hipStream_t)stream));
        HIP_TRY(hipMemsetAsync(d_end, 0, c->nlines * sizeof(uint32_t), (hipStream_t)stream));
        return RRX_OK;
    }
    rc = chunk_index(c, stream);
    if (rc) return rc;
    int e = dev::search_chunks(*ct, c->has_high, c->d_bytes, c->nbytes, c->d_chunk_base, c->nchunks, c->nlines, d_start, d_end, stream);
    if (e) return hip_fail((hipError_t)e, "search_chunks launch");
    return RRX_OK;
}

int rrx_search_all_count(const rrx_regex *re, const rrx_corpus *c, uint32_t *d_count, void *stream) {
    if (!re || !c || (c->nlines && !d_count)) return fail(RRX_ERR_ARG, "null argument");
    const dev::SearchChunkDevice *ct;
    int rc = re->search_tables(c->device, &ct);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->nlines) return RRX_OK;
    if (!ct) {                                                       // accepts "": a match at every offset of the line, its end included
        rc = line_offsets(c, stream);
        if (rc) return rc;
        int e = dev::empty_matches(c->d_line_off, c->nlines, d_count, nullptr, nullptr, nullptr, stream);
        if (e) return hip_fail((hipError_t)e, "empty_matches launch");
        return RRX_OK;
    }
    rc = chunk_index(c, stream);
    if (rc) return rc;
    int e = dev::search_chunks_count(*ct, c->has_high, c->d_bytes, c->nbytes, c->d_chunk_base, c->nchunks, d_count, stream);
    if (e) return hip_fail((hipError_t)e, "search_chunks_count launch");
    return RRX_OK;
}

int rrx_search_all_fill(const rrx_regex *re, const rrx_corpus *c, const uint64_t *d_first, uint32_t *d_start, uint32_t *d_end, void *stream) {
    if (!re || !c || (c->nlines && (!d_first || !d_start || !d_end))) return fail(RRX_ERR_ARG, "null argument");
    const dev::SearchChunkDevice *ct;
    int rc = re->search_tables(c->device, &ct);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->nlines) return RRX_OK;
    if (!ct) {
        rc = line_offsets(c, stream);
        if (rc) return rc;
        int e = dev::empty_matches(c->d_line_off, c->nlines, nullptr, d_first, d_start, d_end, stream);
        if (e) return hip_fail((hipError_t)e, "empty_matches launch");
        return RRX_OK;
    }
    rc = chunk_index(c, stream);
    if (rc) return rc;
    int e = dev::search_chunks_fill(*ct, c->has_high, c->d_bytes, c->nbytes, c->d_chunk_base, c->nchunks, d_first, d_start, d_end, stream);
    if (e) return hip_fail((hipError_t)e, "search_chunks_fill launch");
    return RRX_OK;
}

// count + fill in one call: one launch (decoupled look-back over the chunks' match counts).  A pattern that accepts the empty
// string: the line lengths, a device scan, a fill.
int rrx_search_all(const rrx_regex *re, const rrx_corpus *c, uint64_t *d_first, uint32_t *d_start, uint32_t *d_end, size_t cap, size_t *total,
                   void *stream) {
    if (!re || !c || !total || !d_first || (cap && (!d_start || !d_end))) return fail(RRX_ERR_ARG, "null argument");
    *total = 0;
    const dev::SearchChunkDevice *ct;
    int rc = re->search_tables(c->device, &ct);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    if (!c->nlines) { HIP_TRY(hipMemsetAsync(d_first, 0, sizeof(uint64_t), st)); HIP_TRY(hipStreamSynchronize(st)); return RRX_OK; }
    if (ct) {
        rc = chunk_index(c, stream);
        if (rc) return rc;
        const size_t sb = dev::search_all_scratch_bytes(c->nchunks);
        {
            std::lock_guard<std::mutex> lock(c->mu);
            if (!c->d_all_scratch) HIP_TRY(hipMalloc(&c->d_all_scratch, sb));
        }
        HIP_TRY(hipMemsetAsync(c->d_all_scratch, 0, sb, st));
        int e = dev::search_chunks_all(*ct, c->has_high, c->d_bytes, c->nbytes, c->d_chunk_base, c->nchunks, c->nlines, d_first, d_start, d_end, cap,
                                       c->d_all_scratch, stream);
        if (e) return hip_fail((hipError_t)e, "search_chunks_all launch");
        uint64_t tail[2] = {0, 0};                                 // total, {ticket, error flag}
        HIP_TRY(hipMemcpyAsync(tail, static_cast<uint8_t *>(c->d_all_scratch) + c->nchunks * sizeof(uint64_t), sizeof tail, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (tail[1] >> 32) return fail(RRX_ERR_HIP, "search_all: a chunk's match count was never published (look-back gave up)");
        *total = (size_t)tail[0];
        return RRX_OK;
    }
    // accepts "": count (line length + 1), scan on the device, fill
    rc = line_offsets(c, stream);
    if (rc) return rc;
    uint32_t *d_count = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_count), (c->nlines + 1) * sizeof(uint32_t)));
    uint64_t *d_sums = nullptr;
    hipError_t he = hipMalloc(reinterpret_cast<void **>(&d_sums), dev::scan_scratch_words(c->nlines) * sizeof(uint64_t));
    if (he != hipSuccess) { (void)hipFree(d_count); return hip_fail(he, "hipMalloc"); }
    auto done = [&](int code) { (void)hipFree(d_count); (void)hipFree(d_sums); return code; };
    int e = dev::empty_matches(c->d_line_off, c->nlines, d_count, nullptr, nullptr, nullptr, stream);
    if (!e) e = dev::scan_counts(d_count, d_first, d_sums, c->nlines, stream);  // d_first[nlines] = total
    if (e) return done(hip_fail((hipError_t)e, "empty_matches / scan launch"));
    he = hipMemsetAsync(d_first, 0, sizeof(uint64_t), st);                      // the scan marks entry 0 as a stripe start: not here
    uint64_t tot = 0;
    if (he == hipSuccess) he = hipMemcpyAsync(&tot, d_first + c->nlines, sizeof tot, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    if (he != hipSuccess) return done(hip_fail(he, "search_all scan"));
    *total = (size_t)tot;
    if (tot && cap) {                                                            // matches beyond `cap` are counted, not written (as rrx.h says)
        e = dev::empty_matches(c->d_line_off, c->nlines, nullptr, d_first, d_start, d_end, stream, cap);
        if (e) rc = hip_fail((hipError_t)e, "empty_matches launch");
        if (!rc) { he = hipStreamSynchronize(st); if (he != hipSuccess) rc = hip_fail(he, "search_all fill"); }
    }
    return done(rc);
}

int rrx_bitmap_to_bytes(int device, const uint32_t *d_bits, size_t nlines, uint8_t *d_accept, void *stream) {
    if (nlines && (!d_bits || !d_accept)) return fail(RRX_ERR_ARG, "null argument");
    if (reinterpret_cast<uintptr_t>(d_accept) & 15) return fail(RRX_ERR_ARG, "byte buffer must be 16-byte aligned");
    HIP_TRY(hipSetDevice(device));
    int e = dev::expand_bits(d_bits, nlines, d_accept, stream);
    if (e) return hip_fail((hipError_t)e, "expand_bits launch");
    return RRX_OK;
}

// below these a batch stays on the lane-per-item kernel (the index costs more than it saves)
static constexpr size_t kItemsStripesMin = (size_t)1 << 16;
static constexpr size_t kItemsStripesMinBytes = (size_t)8 << 20;
// a lane (lane group, workgroup) per item
static int match_extents_lanes(const rrx_regex *re, const DeviceTables *t, const uint8_t *b, const uint64_t *d_off, size_t nitems, uint32_t trim,
                               uint8_t *d_accept, void *stream, const uint32_t *only_if = nullptr) {
    int e = re->engine == RRX_ENGINE_NFA_SPARSE ? dev::match_extents_sparse_nfa(t->block, b, d_off, nitems, trim, d_accept, stream)
            : re->engine == RRX_ENGINE_NFA_BLOCK ? dev::match_extents_wave_nfa(t->block, b, d_off, nitems, trim, d_accept, stream)
            : re->engine == RRX_ENGINE_NFA_WAVE ? dev::match_extents_group_nfa(t->group, b, d_off, nitems, trim, d_accept, stream)
            : re->engine == RRX_ENGINE_NFA ? dev::match_extents_nfa(t->nfa, b, d_off, nitems, trim, d_accept, stream)
                                         : dev::match_extents_dfa(t->dfa, b, d_off, nitems, trim, d_accept, stream, only_if);
    if (e) return hip_fail((hipError_t)e, "match_extents launch");
    return RRX_OK;
}
int rrx_match_extents(const rrx_regex *re, int device, const void *d_bytes, const uint64_t *d_off, size_t nitems, uint32_t trim,
                      uint8_t *d_accept, void *stream) {
    if (!re || (nitems && (!d_off || !d_accept))) return fail(RRX_ERR_ARG, "null argument");
    const DeviceTables *t;
    int rc = re->tables(device, &t);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    const uint8_t *b = static_cast<const uint8_t *>(d_bytes);
    // A large batch on a table engine runs stripe-wise over the byte buffer, the item ends taken from a bitmap built from
    // the offsets (kernels_table.hip: match_items_stripes_kernel) and the table a copy of the plain one with an END OF ITEM
    // column.  Needs: trim 0 or 1, at most 126 table states, 16-byte alignment, no item without a byte to carry its mark.
    // ASYNCHRONOUS: nothing is read back.  The host knows neither off[0] nor off[nitems]; it sizes the index for the most
    // the batch can span - what is left of the allocation that holds d_bytes - and the kernels take the real extent from the
    // offsets.  Whether the batch is fit (alignment, no degenerate item, large enough) is decided on the device: the
    // stripe-wise kernel does nothing on an unfit batch and the lane-per-item kernel queued behind it does nothing on a fit one.
    // (r4) trim 1 on a regex with a stride-2 table: the stride-2 items table (it also serves automata whose byte-stride items table
    // is beyond the LDS - a{1,300}: 302 rows of 130 columns)
    const dev::LineDfaDevice *items1 = nullptr;
    const dev::Dfa2Device *items2 = nullptr;
    if (re->engine == RRX_ENGINE_DFA && trim <= 1 && nitems >= kItemsStripesMin && !(reinterpret_cast<uintptr_t>(d_accept) & 15)) {
        if (trim == 1 && re->items_stride2.load()) items2 = re->items2_table(device);
        if (!items2) items1 = re->items_table(device);
    }
    const bool items = items1 || items2;
    size_t bound = 0;
    if (items) {
        hipDeviceptr_t abase = nullptr;
        size_t asize = 0;
        if (hipMemGetAddressRange(&abase, &asize, const_cast<void *>(d_bytes)) == hipSuccess && abase)
            bound = (size_t)(static_cast<const uint8_t *>(abase) + asize - b);
        else (void)hipGetLastError();
    }
    // The tail of the allocation is only a BOUND: a batch carved out of a memory pool (a caching allocator's block, a slice of a
    // column store) would size the index, the stripe and the grids for all of the pool behind it - a 24 MiB batch 6 GiB into a
    // 10 GiB pool: 512 MiB of scratch and two workgroups' worth of stripes.  So the bound is trusted only while it is plausible
    // for the batch: at most 128 bytes per item (string columns; 16 MiB at least).  Beyond that - and for memory whose range
    // the runtime does not report (pools, managed and virtual memory: bound 0) - the batch's real extent is read back, one
    // synchronisation on `stream`, as round 2 did for every batch.  (The kernels take the extent from the offsets either way.)
    if (items) {
        const size_t plausible = std::max<size_t>(nitems * 128, (size_t)16 << 20);
        if (!bound || bound > plausible) {
            uint64_t first = 0, last = 0;
            HIP_TRY(hipMemcpyAsync(&first, d_off, sizeof first, hipMemcpyDeviceToHost, (hipStream_t)stream));
            HIP_TRY(hipMemcpyAsync(&last, d_off + nitems, sizeof last, hipMemcpyDeviceToHost, (hipStream_t)stream));
            HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            const size_t extent = last > first ? (size_t)last : 0;
            bound = bound ? std::min(bound, extent) : extent;
        }
    }
    if (items && bound >= kItemsStripesMinBytes) {
        hipStream_t st = (hipStream_t)stream;
        std::lock_guard<std::mutex> lock(re->onepass_mu);
        void *buf = nullptr;
        const size_t ib = dev::items_index_bytes(bound, nitems);
        rc = re->onepass_for(device, ib + dev::items_result_bytes(nitems), &buf, st);      // (ordered behind the scratch's last user)
        if (rc) return rc;
        uint32_t *d_flag = nullptr;
        int le = dev::items_index_build(bound, d_off, nitems, trim, buf, &d_flag, stream, b, kItemsStripesMinBytes);
        if (!le) le = items2 ? dev::items_match2(*items2, b, bound, nitems, buf, static_cast<uint8_t *>(buf) + ib, d_accept, stream, d_off, d_flag)
                             : dev::items_match(*items1, b, bound, nitems, trim, buf, static_cast<uint8_t *>(buf) + ib, d_accept, stream, d_off, d_flag);
        if (!le) rc = match_extents_lanes(re, t, b, d_off, nitems, trim, d_accept, stream, d_flag);
        const int rc2 = re->onepass_done(device, st);            // (whatever was queued: the next user waits for it)
        if (le) return hip_fail((hipError_t)le, "match_items_stripes launch");
        return rc ? rc : rc2;
    }
    return match_extents_lanes(re, t, b, d_off, nitems, trim, d_accept, stream);
}

// A batch of items indexed once (item-end bitmap + stripe base), matched by many patterns: rrx_corpus' counterpart for an
// offsets array.  stripes = false: the batch does not admit the stripe-wise kernel (trim > 1, an empty item at trim 0,
// alignment); rrx_match_items then runs the lane-per-item kernel.
struct rrx_items {
    int device = 0;
    const uint8_t *d_bytes = nullptr;
    const uint64_t *d_off = nullptr;
    size_t nitems = 0, nbytes = 0;       // nbytes = off[nitems] - off[0]
    uint64_t first = 0;
    uint32_t trim = 0;
    bool stripes = false;
    void *d_index = nullptr;
    mutable std::mutex mu;
    mutable void *d_result = nullptr;    // result bitmap of a match (one match at a time per handle)
};
int rrx_items_create(int device, const void *d_bytes, const uint64_t *d_off, size_t nitems, uint32_t trim, void *stream, rrx_items **out) {
    if (!out || (nitems && (!d_bytes || !d_off))) return fail(RRX_ERR_ARG, "null argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    rrx_items *it = new rrx_items();
    it->device = device; it->d_bytes = static_cast<const uint8_t *>(d_bytes); it->d_off = d_off; it->nitems = nitems; it->trim = trim;
    hipStream_t st = (hipStream_t)stream;
    if (nitems && trim <= 1) {
        uint64_t first = 0, last = 0;
        hipError_t e = hipMemcpyAsync(&first, d_off, sizeof first, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(&last, d_off + nitems, sizeof last, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { delete it; return hip_fail(e, "items offsets readback"); }
        it->first = first;
        if (last > first && !(reinterpret_cast<uintptr_t>(it->d_bytes + first) & 15)) {
            it->nbytes = (size_t)(last - first);
            e = hipMalloc(&it->d_index, dev::items_index_bytes(it->nbytes, nitems));
            if (e == hipSuccess) e = hipMalloc(&it->d_result, dev::items_result_bytes(nitems));
            if (e != hipSuccess) { rrx_items_free(it); return hip_fail(e, "hipMalloc(items index)"); }
            uint32_t *d_flag = nullptr;
            int le = dev::items_index_build(it->nbytes, d_off, nitems, trim, it->d_index, &d_flag, stream);
            uint32_t degenerate = 1;
            if (!le) { e = hipMemcpyAsync(&degenerate, d_flag, sizeof degenerate, hipMemcpyDeviceToHost, st); if (e == hipSuccess) e = hipStreamSynchronize(st); }
            if (le || e != hipSuccess) { rrx_items_free(it); return le ? hip_fail((hipError_t)le, "items index launch") : hip_fail(e, "items index"); }
            it->stripes = degenerate == 0;
        }
    }
    *out = it;
    return RRX_OK;
}
size_t rrx_items_count(const rrx_items *it) { return it ? it->nitems : 0; }
int rrx_items_stripe_wise(const rrx_items *it) { return it && it->stripes ? 1 : 0; }
void rrx_items_free(rrx_items *it) {
    if (!it) return;
    (void)hipSetDevice(it->device);
    if (it->d_index) (void)hipFree(it->d_index);
    if (it->d_result) (void)hipFree(it->d_result);
    delete it;
}
int rrx_match_items(const rrx_regex *re, const rrx_items *it, uint8_t *d_accept, void *stream) {
    if (!re || !it || (it->nitems && !d_accept)) return fail(RRX_ERR_ARG, "null argument");
    if (!it->nitems) return RRX_OK;
    const DeviceTables *t;
    int rc = re->tables(it->device, &t);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(it->device));
    if (it->stripes && re->engine == RRX_ENGINE_DFA && !(reinterpret_cast<uintptr_t>(d_accept) & 15)) {
        const dev::Dfa2Device *items2 = (it->trim == 1 && re->items_stride2.load()) ? re->items2_table(it->device) : nullptr;
        const dev::LineDfaDevice *items1 = items2 ? nullptr : re->items_table(it->device);
        if (items1 || items2) {
            std::lock_guard<std::mutex> lock(it->mu);
            int le = items2 ? dev::items_match2(*items2, it->d_bytes + it->first, it->nbytes, it->nitems, it->d_index, it->d_result, d_accept, stream)
                            : dev::items_match(*items1, it->d_bytes + it->first, it->nbytes, it->nitems, it->trim, it->d_index, it->d_result, d_accept, stream);
            if (le) return hip_fail((hipError_t)le, "match_items launch");
            return RRX_OK;
        }
    }
    // the batch or the pattern does not admit the stripe-wise kernel (the index said so once: no second attempt)
    return match_extents_lanes(re, t, it->d_bytes, it->d_off, it->nitems, it->trim, d_accept, stream);
}

// One device-resident string of any length.  Long strings take the chunk-map path when the automaton has a small
// table (every chunk stepped from every state, maps composed); the rest is one item of the extents kernel.
// `scratch`/`scratch_bytes`: caller-provided device memory (rrx_match_cstr passes the tail of its own buffer).
static int match_string_with(const rrx_regex *re, int device, const DeviceTables *t, const uint8_t *d_bytes, size_t nbytes, uint8_t *d_accept,
                             uint8_t *scratch, hipStream_t st) {
    const bool table_engine = re->engine == RRX_ENGINE_DFA;
    if (table_engine && nbytes >= kLongStringBytesTable && t->dfa.nstates && t->dfa.nstates <= dev::kLongMaxStates) {
        uint32_t chunk = 0;
        (void)dev::long_scratch_bytes(t->dfa.nstates, nbytes, &chunk);
        int le = dev::match_long_dfa(t->dfa, d_bytes, nbytes, chunk, scratch, d_accept, st);
        if (le) return hip_fail((hipError_t)le, "match_long launch");
        return RRX_OK;
    }
    if (re->engine == RRX_ENGINE_NFA && nbytes >= kLongStringBytes && t->nfa.nbits <= kLongNfaMaxBits) {
        uint32_t chunk = 0, nchunks = 0;
        (void)dev::long_nfa_scratch_bytes(t->nfa, nbytes, &chunk, &nchunks);
        int le = dev::match_long_nfa(t->nfa, d_bytes, nbytes, chunk, nchunks, scratch, d_accept, st);
        if (le) return hip_fail((hipError_t)le, "match_long_nfa launch");
        return RRX_OK;
    }
    const uint64_t off[2] = {0, nbytes};
    uint64_t *d_off = reinterpret_cast<uint64_t *>(scratch);
    hipError_t e = hipMemcpyAsync(d_off, off, sizeof off, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);                    // `off` leaves scope
    if (e != hipSuccess) return hip_fail(e, "extent upload");
    return rrx_match_extents(re, device, d_bytes, d_off, 1, 0, d_accept, st);
}
static size_t match_string_scratch_bytes(const rrx_regex *re, const DeviceTables *t, size_t nbytes) {
    const bool table_engine = re->engine == RRX_ENGINE_DFA;
    if (table_engine && nbytes >= kLongStringBytesTable && t->dfa.nstates && t->dfa.nstates <= dev::kLongMaxStates) {
        uint32_t chunk = 0;
        return dev::long_scratch_bytes(t->dfa.nstates, nbytes, &chunk);
    }
    if (re->engine == RRX_ENGINE_NFA && nbytes >= kLongStringBytes && t->nfa.nbits <= kLongNfaMaxBits) {
        uint32_t chunk = 0, nchunks = 0;
        return dev::long_nfa_scratch_bytes(t->nfa, nbytes, &chunk, &nchunks);
    }
    return 2 * sizeof(uint64_t);
}

int rrx_match_string(const rrx_regex *re, int device, const void *d_bytes, size_t nbytes, uint8_t *d_accept, void *stream) {
    if (!re || (nbytes && !d_bytes) || !d_accept) return fail(RRX_ERR_ARG, "null argument");
    const DeviceTables *t;
    int rc = re->tables(device, &t);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(re->scratch_mu);
    void *scratch = nullptr;
    rc = re->scratch_for(device, match_string_scratch_bytes(re, t, nbytes), &scratch);
    if (rc) return rc;
    rc = match_string_with(re, device, t, static_cast<const uint8_t *>(d_bytes), nbytes, d_accept, static_cast<uint8_t *>(scratch),
                           static_cast<hipStream_t>(stream));
    const hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream));      // the scratch is reused by the next call
    if (!rc && e != hipSuccess) rc = hip_fail(e, "match_string");
    return rc;
}

// Host buffer in, one byte per string out.  Large inputs are cut into line-aligned chunks; the upload of chunk i+1 is
// queued before index + match + download of chunk i (two device buffers).  PCIe inclusive; never the benchmarked rate.
static int match_host_chunk(const rrx_regex *re, int device, uint8_t *d_text, size_t len, hipStream_t st, uint32_t *d_bits,
                            uint8_t *d_acc, uint8_t *accept, size_t cap, size_t line_off, size_t *nlines_out) {
    rrx_corpus *c = nullptr;
    int rc = rrx_corpus_create(device, d_text, len, st, &c);             // waits for the chunk's copy and index
    if (rc) return rc;
    const size_t n = c->nlines;
    rc = rrx_match_corpus(re, c, d_bits, st);
    if (!rc) rc = rrx_bitmap_to_bytes(device, d_bits, n, d_acc, st);
    if (!rc && line_off < cap) {
        const size_t take = n < cap - line_off ? n : cap - line_off;
        hipError_t e = hipMemcpyAsync(accept + line_off, d_acc, take, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) rc = hip_fail(e, "accept readback");
    }
    hipError_t e = hipStreamSynchronize(st);                             // the index arrays of `c` are freed next
    if (!rc && e != hipSuccess) rc = hip_fail(e, "chunk sync");
    rrx_corpus_free(c);
    *nlines_out = n;
    return rc;
}

int rrx_match_host(const rrx_regex *re, int device, const void *bytes, size_t nbytes, uint8_t *accept, size_t cap, size_t *nlines) {
    if (!re || (nbytes && !bytes) || !nlines) return fail(RRX_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(device));
    *nlines = 0;
    if (!nbytes) return RRX_OK;
    const uint8_t *host = static_cast<const uint8_t *>(bytes);
    const size_t kChunk = (size_t)256 << 20;
    const size_t buf_bytes = (nbytes < kChunk ? nbytes : kChunk) + 64;
    const size_t max_lines = buf_bytes;                                   // a chunk of n bytes holds at most n lines
    uint8_t *d_text[2] = {nullptr, nullptr}, *d_acc[2] = {nullptr, nullptr};
    uint32_t *d_bits[2] = {nullptr, nullptr};
    hipStream_t st[2] = {nullptr, nullptr};
    const int nbuf = nbytes > kChunk ? 2 : 1;
    int rc = RRX_OK;
    hipError_t e = hipSuccess;
    for (int i = 0; i < nbuf && e == hipSuccess; i++) {
        e = hipMalloc(reinterpret_cast<void **>(&d_text[i]), buf_bytes);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_acc[i]), max_lines + 64);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_bits[i]), (max_lines / 32 + 4) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipStreamCreate(&st[i]);
    }
    if (e != hipSuccess) rc = hip_fail(e, "pipeline buffers");

    // chunk boundaries: right after the last '\n' of each window (a window without any '\n' is taken whole: the
    // line continues, and since a line must be matched by one launch such inputs fall back to one big chunk)
    std::vector<size_t> cuts{0};
    while (!rc && cuts.back() < nbytes) {
        size_t lo = cuts.back(), hi = lo + kChunk < nbytes ? lo + kChunk : nbytes;
        if (hi < nbytes) {
            size_t q = hi;
            while (q > lo && host[q - 1] != '\n') q--;
            if (q == lo) { rc = fail(RRX_ERR_UNSUPPORTED, "a single line longer than 256 MiB: use rrx_corpus_create on a device buffer"); break; }
            hi = q;
        }
        cuts.push_back(hi);
    }
    const size_t nchunks = cuts.size() - 1;
    size_t line_off = 0;
    // The caller's pages are NOT pinned: on this platform the runtime's own staged copy from pageable memory runs at
    // 49 GB/s (57 pinned), while pinning costs as much as the copy (hipHostRegister + hipHostUnregister: ~30 ms per
    // GiB).  Measured on 4 GiB: 42 GB/s unpinned, 31 GB/s pinning everything first, 33 GB/s pinning 64-MiB windows on
    // a helper thread ahead of the uploads.
    if (!rc) {
        e = hipMemcpyAsync(d_text[0], host, cuts[1] - cuts[0], hipMemcpyHostToDevice, st[0]);
        if (e != hipSuccess) rc = hip_fail(e, "chunk upload");
    }
    for (size_t i = 0; i < nchunks && !rc; i++) {
        const int cur = (int)(i & 1) % nbuf, nxt = (int)((i + 1) & 1) % nbuf;
        if (i + 1 < nchunks) {                                             // next chunk's upload is queued before this chunk's work
            e = hipMemcpyAsync(d_text[nxt], host + cuts[i + 1], cuts[i + 2] - cuts[i + 1], hipMemcpyHostToDevice, st[nxt]);
            if (e != hipSuccess) { rc = hip_fail(e, "chunk upload"); break; }
        }
        size_t n = 0;
        rc = match_host_chunk(re, device, d_text[cur], cuts[i + 1] - cuts[i], st[cur], d_bits[cur], d_acc[cur], accept, cap, line_off, &n);
        line_off += n;
    }
    for (int i = 0; i < nbuf; i++) if (st[i]) (void)hipStreamSynchronize(st[i]);
    for (int i = 0; i < 2; i++) {
        if (d_text[i]) (void)hipFree(d_text[i]);
        if (d_acc[i]) (void)hipFree(d_acc[i]);
        if (d_bits[i]) (void)hipFree(d_bits[i]);
        if (st[i]) (void)hipStreamDestroy(st[i]);
    }
    *nlines = line_off;
    return rc;
}

int rrx_match_cstr(const rrx_regex *re, int device, const char *text, int *accepted, size_t *len) {
    if (!re || !text || !accepted) return fail(RRX_ERR_ARG, "null argument");
    const size_t n = std::strlen(text);                       // regex.h:157: consume up to the terminator
    if (len) *len = n;
    const DeviceTables *t;
    int rc = re->tables(device, &t);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    // one persistent device buffer: [text, padded to 16 | accept byte, padded to 16 | scratch of the match]
    std::lock_guard<std::mutex> lock(re->scratch_mu);
    const size_t acc_at = (n + 15) & ~(size_t)15, scratch_at = acc_at + 16;
    void *buf = nullptr;
    rc = re->scratch_for(device, scratch_at + match_string_scratch_bytes(re, t, n), &buf);
    if (rc) return rc;
    uint8_t *d = static_cast<uint8_t *>(buf);
    hipError_t e = n ? hipMemcpy(d, text, n, hipMemcpyHostToDevice) : hipSuccess;
    if (e != hipSuccess) return hip_fail(e, "text upload");
    rc = match_string_with(re, device, t, d, n, d + acc_at, d + scratch_at, nullptr);
    uint8_t a = 0;
    if (!rc) { e = hipMemcpy(&a, d + acc_at, 1, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = hip_fail(e, "accept readback"); }
    *accepted = a;
    return rc;
}

}  // extern "C"
