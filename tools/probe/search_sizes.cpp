// Sizes of the search tables (host only): rows x columns of the line-mode product table, what a Mealy minimisation of it
// leaves, and the number of distinct pair columns of its stride-2 form.
// build + run: g++ -std=c++17 -O2 -o /tmp/search_sizes tools/probe/search_sizes.cpp roaringregex_amd/csrc/frontend.cpp roaringregex_amd/csrc/lower.cpp -pthread && /tmp/search_sizes
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../roaringregex_amd/csrc/frontend.hpp"
#include "../../roaringregex_amd/csrc/lower.hpp"

using namespace rrx;

static uint32_t mealy_classes(const SearchLineProgram &s, std::vector<uint32_t> &block) {
    const uint32_t R = s.nrows, C = s.ncols;
    block.assign(R, 0);
    uint32_t nb = 1;
    for (;;) {
        std::map<std::vector<uint32_t>, uint32_t> sig;
        std::vector<uint32_t> nbk(R);
        for (uint32_t r = 0; r < R; r++) {
            std::vector<uint32_t> v(C + 1);
            v[0] = block[r];
            for (uint32_t c = 0; c < C; c++) { const uint32_t e = s.table[(size_t)r * C + c]; v[c + 1] = (e & 0xffff0000u) | block[e & 0xffffu]; }
            auto it = sig.emplace(v, (uint32_t)sig.size()).first;
            nbk[r] = it->second;
        }
        const uint32_t n2 = (uint32_t)sig.size();
        block = nbk;
        if (n2 == nb) break;
        nb = n2;
    }
    return nb;
}

static uint32_t pair_columns(const SearchLineProgram &s, const std::vector<uint32_t> &block, uint32_t nb, bool restart) {
    const uint32_t C = s.ncols;
    // representative row per block
    std::vector<uint32_t> rep(nb, UINT32_MAX);
    for (uint32_t r = 0; r < s.nrows; r++) if (rep[block[r]] == UINT32_MAX) rep[block[r]] = r;
    std::map<std::vector<uint32_t>, uint32_t> cols;
    for (uint32_t a = 0; a < C; a++)
        for (uint32_t b = 0; b < C; b++) {
            std::vector<uint32_t> col(nb);
            for (uint32_t k = 0; k < nb; k++) {
                uint32_t e1 = s.table[(size_t)rep[k] * C + a];
                uint32_t r1 = e1 & 0xffffu;
                if (restart && (e1 & kSearchHit)) r1 = s.start;
                uint32_t e2 = s.table[(size_t)r1 * C + b];
                uint32_t r2 = e2 & 0xffffu;
                if (restart && (e2 & kSearchHit)) r2 = s.start;
                col[k] = block[r2] | (e1 >> 16) << 16 | (e2 >> 16) << 20;
            }
            cols.emplace(col, (uint32_t)cols.size());
        }
    return (uint32_t)cols.size();
}

int main(int argc, char **argv) {
    std::string kw, kw1000;
    for (int i = 1; i <= 1000; i++) kw1000 += (i > 1 ? "|k" : "k") + std::to_string(i);
    std::vector<std::pair<std::string, std::string>> pats = {
        {"email", "[A-Za-z0-9._]+@[A-Za-z0-9.]+"},
        {"U2", "(http|https|ftp)://([a-z0-9-]{1,16}\\.){1,3}[a-z]{2,6}(:[0-9]{1,5})?(/[A-Za-z0-9._~%-]*)*(\\?[A-Za-z0-9._~%=&-]*)?(#[A-Za-z0-9._~%-]*)?"},
        {"a{1,300}", "a{1,300}"},
        {"k1000", kw1000},
        {"k1000c", ".*(" + kw1000 + ").*"},
        {"abx", "[ab]*a[ab]{11}x"},
        {"abc", "abc"},
    };
    for (int i = 1; i < argc; i++) pats.push_back({argv[i], argv[i]});
    for (auto &pp : pats) {
        const RefAutomaton ref = build_reference_automaton(pp.second);
        const Reduced red = reduce(trim(ref));
        DfaProgram f, r, anch;
        if (!search_dfas(red, 16384, f, r) || !lower_dfa(red, 16384, anch)) { std::printf("%-10s search DFAs too large\n", pp.first.c_str()); continue; }
        SearchLineProgram s;
        if (!lower_search_line(f, &anch, 16383, s)) { std::printf("%-10s fwd %u rev %u anch %u: no line table\n", pp.first.c_str(), f.nstates, r.nstates, anch.nstates); continue; }
        std::vector<uint32_t> block;
        const uint32_t nb = mealy_classes(s, block);
        std::vector<uint32_t> ident(s.nrows);
        for (uint32_t i = 0; i < s.nrows; i++) ident[i] = i;
        const uint32_t pc = pair_columns(s, block, nb, false), pca = pair_columns(s, block, nb, true);
        std::printf("%-10s fwd %u rev %u anch %u ncls %u | line table %u x %u = %zu B | minimised rows %u (%zu B) | pair columns first %u all %u | T2 4B: %zu / %zu B, 2B: %zu B\n",
                    pp.first.c_str(), f.nstates, r.nstates, anch.nstates, f.ncls, s.nrows, s.ncols, (size_t)s.nrows * s.ncols * 4, nb, (size_t)nb * s.ncols * 4, pc, pca,
                    (size_t)nb * pc * 4, (size_t)nb * pca * 4, (size_t)nb * pc * 2);
    }
    return 0;
}
