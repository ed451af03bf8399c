// The background order search's state machine (lower.hpp: TableOrderSearch) under ThreadSanitizer (CPU only; GPU sanitizers
// are not available on this pool): several threads race to start the search of one regex (rrx_match_corpus's first call and
// rrx_order_table), others poll its state the way rrx_table_order does and read the "descriptor" under the owner's mutex the way
// a launch does, and the owner is destroyed while the search may still run.  build + run: make -C tools/sanitize tsan
#include <atomic>
#include <cstdio>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "../../roaringregex_amd/csrc/frontend.hpp"
#include "../../roaringregex_amd/csrc/lower.hpp"

using namespace rrx;

struct Owner {                       // what rrx_regex keeps around the search
    Dfa2Program dfa2;
    std::mutex mu;
    std::vector<uint32_t> row_slot, col_slot;
    Dfa2OrderStats stats;
    TableOrderSearch search;         // (last member: destroyed - and so joined - first)
};

int main() {
    const RefAutomaton ref = build_reference_automaton(
        "(http|https|ftp)://([a-z0-9-]{1,16}\\.){1,3}[a-z]{2,6}(:[0-9]{1,5})?(/[A-Za-z0-9._~%-]*)*(\\?[A-Za-z0-9._~%=&-]*)?(#[A-Za-z0-9._~%-]*)?");
    const Reduced red = reduce(trim(ref));
    DfaProgram d;
    Dfa2Program d2;
    if (!lower_dfa(red, 16384, d) || !lower_dfa2(d, 1024, d2)) { std::printf("no stride-2 table\n"); return 1; }
    std::mt19937 rng(7);
    const uint32_t lanes = 64, per = 64;
    std::vector<uint8_t> sample((size_t)lanes * per);
    for (auto &b : sample) { const unsigned r = rng() % 40; b = r == 0 ? '\n' : (uint8_t)("abckx01./:h tp"[r % 14]); }
    int started_total = 0;
    for (int round = 0; round < 40; round++) {
        auto owner = std::make_unique<Owner>();
        owner->dfa2 = d2;
        std::atomic<int> started{0}, applied{0};
        std::atomic<bool> stop{false};
        auto apply = [&, o = owner.get()](std::vector<uint32_t> &&r, std::vector<uint32_t> &&c, const Dfa2OrderStats &st) {
            std::lock_guard<std::mutex> lock(o->mu);
            o->row_slot.swap(r); o->col_slot.swap(c); o->stats = st;
            applied++;
        };
        std::vector<std::thread> th;
        for (int k = 0; k < 3; k++)                                   // racing deciders: background, caller's thread, skip
            th.emplace_back([&, k, o = owner.get()] {
                if (k == 2 && round % 3 == 0) { if (o->search.skip()) started++; return; }
                if (o->search.start(o->dfa2, sample, lanes, per, /*background=*/k == 0, apply)) started++;
            });
        for (int k = 0; k < 3; k++)                                   // pollers / launches
            th.emplace_back([&, o = owner.get()] {
                while (!stop.load()) {
                    const TableOrderSearch::State st = o->search.state();
                    std::lock_guard<std::mutex> lock(o->mu);
                    if (st == TableOrderSearch::kDone && o->row_slot.size() != o->dfa2.nstates) { std::printf("DONE WITHOUT AN ORDER\n"); std::abort(); }
                    if (!o->row_slot.empty() && o->row_slot[0] != 0) { std::printf("DEAD ROW MOVED\n"); std::abort(); }
                }
            });
        for (int k = 0; k < 3; k++) th[k].join();
        if (started.load() != 1) { std::printf("DECIDED %d TIMES\n", started.load()); return 1; }
        if (round % 2) owner->search.wait();                          // else: the destructor meets a running search
        stop = true;
        for (size_t k = 3; k < th.size(); k++) th[k].join();
        const bool skipped = owner->search.state() == TableOrderSearch::kSkipped;
        owner.reset();                                                // joins
        if (!skipped && applied.load() != 1) { std::printf("APPLIED %d TIMES\n", applied.load()); return 1; }
        started_total++;
    }
    std::printf("order search: %d rounds raced, no report\n", started_total);
    return 0;
}
