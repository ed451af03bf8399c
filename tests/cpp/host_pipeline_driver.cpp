// host_pipeline_driver.cpp — runs the host compile pipeline (pattern -> reference-numbered automaton -> trim ->
// reduce -> NFA / DFA / stride-2 programs) over the patterns given on stdin, one per line.  Built by
// tests/test_lowering.py with -fsanitize=address,undefined (sanitizers run on the CPU build only) from
// csrc/frontend.cpp + csrc/lower.cpp: no HIP involved.  Prints one summary line per pattern.
#include <cstdio>
#include <iostream>
#include <string>

#include "frontend.hpp"
#include "lower.hpp"

int main() {
    std::string p;
    size_t n = 0, rejected = 0;
    while (std::getline(std::cin, p)) {
        n++;
        try {
            rrx::RefAutomaton a = rrx::build_reference_automaton(p.c_str());
            rrx::Trimmed t = rrx::trim(a);
            rrx::Reduced r = rrx::reduce(t);
            rrx::NfaProgram nfa, wave;
            rrx::DfaProgram dfa;
            rrx::Dfa2Program dfa2;
            const bool has_nfa = rrx::lower_nfa(r, 512, nfa, true);
            const bool has_wave = rrx::lower_nfa(r, 4096, wave, false);
            const bool has_dfa = rrx::lower_dfa(r, 16384, dfa);
            const bool has_dfa2 = has_dfa && rrx::lower_dfa2(dfa, 1024, dfa2);
            std::printf("%zu ok useful %u nodes %zu nfa %d/%u wave %d dfa %d/%u dfa2 %d/%u\n", n, t.n, r.nodes.size(), (int)has_nfa,
                        has_nfa ? nfa.nbits : 0u, (int)has_wave, (int)has_dfa, has_dfa ? dfa.nstates : 0u, (int)has_dfa2,
                        has_dfa2 ? dfa2.ncols : 0u);
        } catch (const rrx::PatternError &e) {
            rejected++;
            std::printf("%zu rejected %s\n", n, e.what());
        }
    }
    std::printf("done %zu rejected %zu\n", n, rejected);
    return 0;
}
