// Host compile pipeline under AddressSanitizer + UBSan (CPU only; GPU sanitizers are not available on this pool):
// frontend -> trim -> reduce (worklist quotients) -> NFA / DFA / stride-2 lowering -> order_dfa2 on a random text sample.
// build + run: make -C tools/sanitize
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include "../../roaringregex_amd/csrc/frontend.hpp"
#include "../../roaringregex_amd/csrc/lower.hpp"

using namespace rrx;

static std::string random_pattern(std::mt19937 &rng) {
    static const char *atoms[] = {"a", "b", "c", "x", "k", "0", "1", "\\.", ".", "[a-c]", "[^ab]", "[0-9]", "(a|b)", "(ab|c)", "(a|bc|d)"};
    static const char *post[] = {"", "", "", "*", "+", "?", "{2}", "{1,3}", "{2,}", "{1,12}"};
    std::string p;
    const int n = 1 + (int)(rng() % 6);
    for (int i = 0; i < n; i++) {
        std::string a = atoms[rng() % 15];
        if (rng() % 5 == 0) a = "(" + a + atoms[rng() % 15] + ")";
        p += a + post[rng() % 10];
        if (rng() % 7 == 0 && i + 1 < n) p += "|";
    }
    return p;
}

int main() {
    std::mt19937 rng(12345);
    std::vector<std::string> pats = {
        "abc", "a*", "(ab)+", "a{1,85}", "a{1,300}", "(a|b)*a(a|b){40}", "(a|b)*a(a|b){600}", "(a|b)*a(a|b){3000}", "(a|b)*a.{6000}",
        "[A-Za-z0-9._]+@[A-Za-z0-9.]+",
        "(http|https|ftp)://([a-z0-9-]{1,16}\\.){1,3}[a-z]{2,6}(:[0-9]{1,5})?(/[A-Za-z0-9._~%-]*)*(\\?[A-Za-z0-9._~%=&-]*)?(#[A-Za-z0-9._~%-]*)?",
        "((a|b)*a(a|b){700}c|(b|c)*b(b|c){800}a)*", "(ab|ba){1,120}", "b*a{1,440}", "x?y?z?", "(a*b)*",
        // nullable folds the front end keeps as shared row pieces, and what the domination proofs of trim() meet in them
        "a{1,20000}", "[ab]{1,9000}", "(ab){1,5000}", "(abc|de){1,3000}", "(a+b+){1,1500}", "(a{0,3}b{0,3}){1,400}", "([ab]{1,40}c?){1,60}",
        "(a{1,40}){1,40}", "(abcdefghij){1,900}", "((ab){1,20}c){1,30}", "(a?b?c?){1,700}", "((a|b)*c?){1,500}", "(a?|b?c)*{1,300}",
        "((a|b)*a(a|b){5}){1,200}", "(a*b*){1,500}x", "(a?){3000}"};
    std::string kw;
    for (int i = 1; i <= 300; i++) kw += (i > 1 ? "|k" : "k") + std::to_string(i);
    pats.push_back(kw);
    pats.push_back(".*(" + kw + ").*");
    for (int i = 0; i < 400; i++) pats.push_back(random_pattern(rng));
    size_t ok = 0, refused = 0, ordered = 0;
    for (const std::string &p : pats) {
        try {
            const RefAutomaton ref = build_reference_automaton(p);
            const Trimmed t = trim(ref);
            const Reduced red = reduce(t);
            NfaProgram nfa, big;
            (void)lower_nfa(red, 512, nfa, true, true);
            (void)lower_nfa(red, 65536, big, false, true);
            DfaProgram d;
            if (lower_dfa(red, 16384, d)) {
                Dfa2Program d2;
                if (d.nstates <= 4096 && lower_dfa2(d, 1024, d2)) {
                    const uint32_t lanes = 64, per = 128;
                    std::vector<uint8_t> sample((size_t)lanes * per);
                    for (auto &b : sample) { const unsigned r = rng() % 40; b = r == 0 ? '\n' : r < 3 ? (uint8_t)(rng() & 0xff) : (uint8_t)("abckx01./:h tp"[r % 14]); }
                    std::vector<uint32_t> rows, cols;
                    const Dfa2OrderStats st = order_dfa2(d2, sample.data(), lanes, per, rows, cols);
                    if (rows.size() != d2.nstates || cols.size() != d2.ncols || rows[0] != 0 || st.after > st.before + 1e-9) { std::printf("ORDER BROKEN %s\n", p.c_str()); return 1; }
                    std::vector<uint8_t> seen(rows.size(), 0);
                    for (uint32_t r : rows) { if (r >= rows.size() || seen[r]) { std::printf("NOT A PERMUTATION %s\n", p.c_str()); return 1; } seen[r] = 1; }
                    ordered++;
                }
                DfaProgram f, r;
                (void)search_dfas(red, 16384, f, r);
            }
            ok++;
        } catch (const PatternError &) {
            refused++;
        }
    }
    std::printf("host pipeline under ASan/UBSan: %zu patterns lowered, %zu refused by the front end, %zu stride-2 tables ordered\n", ok, refused, ordered);
    return 0;
}
