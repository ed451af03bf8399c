"""The oracle against every known answer the survey recorded from the reference's own code
(SURVEY.md 8(c) -> tests/golden/kat.json).  CPU only."""
import pytest
from pyoracle import OracleRegex, OracleError


def test_kat_accept_reject(kat):
    for k in kat["kat"]:
        r = OracleRegex(k["pattern"])
        if k["states_n"] is not None:
            assert r.states_n == k["states_n"], k["pattern"]
        for t in k["accepts"]:
            assert r.accepts(t), (k["pattern"], t)
        for t in k["rejects"]:
            assert not r.accepts(t), (k["pattern"], t)


def test_class_dispatch_thresholds(kat):
    # Parser.cpp:165-168
    for k in kat["kat"]:
        if k["states_n"] is None:
            continue
        n = k["states_n"]
        want = 0 if n > 256 else 4 if n > 128 else 2 if n > 64 else 1
        assert OracleRegex(k["pattern"]).set_class == want


def test_big_state_counts(kat):
    for k in kat["big_states"]:
        r = OracleRegex(k["pattern"])
        assert r.states_n == k["states_n"]
        assert r.set_class == 0


def test_intended_semantics_where_reference_is_broken(kat):
    # >256 states: the reference aliases states mod 256 (NFA.cc:10-11); the oracle must give the intended
    # answer, which differs from what the broken reference printed (SURVEY.md 8(c) divergence record).
    for k in kat["broken_reference"]:
        r = OracleRegex(k["pattern"])
        assert r.accepts(k["text"]) == k["intended"]
        assert k["intended"] != k["reference"]


def _stats(r):
    n = r.states_n
    rows = {(s, c): r.row(s, c) for s in range(n) for c in range(128)}
    fin = set(r.finals())
    reach, st = {r.initial}, [r.initial]
    while st:
        s = st.pop()
        for c in range(128):
            for t in rows[(s, c)]:
                if t not in reach:
                    reach.add(t)
                    st.append(t)
    pred = {}
    for (s, c), ts in rows.items():
        for t in ts:
            pred.setdefault(t, set()).add(s)
    co, st = set(fin), list(fin)
    while st:
        t = st.pop()
        for s in pred.get(t, ()):
            if s not in co:
                co.add(s)
                st.append(s)
    cols = {tuple(tuple(rows[(s, c)]) for s in range(n)) for c in range(128)}
    return len(reach), len(reach & co), len(cols), max(len(v) for v in rows.values())


def test_table_statistics_match_reference_dumps(kat):
    # measured from the reference's print() dumps (NFA.cc:14-41)
    for k in kat["table_stats"]:
        r = OracleRegex(k["pattern"])
        assert r.states_n == k["states_n"]
        reach, useful, ncls, maxpop = _stats(r)
        assert reach == k["reachable"], k["pattern"]
        assert useful == k["useful"], k["pattern"]
        assert ncls == k["byte_classes"], k["pattern"]
        if k["max_row_popcount"] is not None:
            assert maxpop == k["max_row_popcount"], k["pattern"]


def test_backward_table_is_transpose():
    # NFA.cc:52-53,63-64,118: bwd[t][c] contains s  <=>  fwd[s][c] contains t
    for p in ["abc", "(a|b)*c", "a(b|c)?d", "[a-c]{2,4}x", "(ab|cd)+e?"]:
        r = OracleRegex(p)
        n = r.states_n
        for c in range(128):
            f = {(s, t) for s in range(n) for t in r.row(s, c, True)}
            b = {(s, t) for t in range(n) for s in r.row(t, c, False)}
            assert f == b, (p, c)


@pytest.mark.parametrize("p", ["a)", "|a", "a|", "()", "*a", "?", "+", "a{2", "[", "a{", "caf\xe9"])
def test_patterns_with_undefined_behaviour_in_the_reference_are_errors(p):
    with pytest.raises(OracleError):
        OracleRegex(p)


@pytest.mark.parametrize("p,equiv", [("(a", "a"), ("[abc", "[ab]"), ("a\\", "a"), ("a{}", "a"), ("a{2,3", "a{2,3}"), ("[a-", "[a]")])
def test_silently_accepted_malformed_patterns(p, equiv):
    # SURVEY.md 8(b) Errors: "[abc", "(a" compile silently
    a, b = OracleRegex(p), OracleRegex(equiv)
    for t in ["", "a", "b", "c", "aa", "aaa", "ab", "abc", "aaaa"]:
        assert a.accepts(t) == b.accepts(t), (p, t)


def test_empty_pattern_is_a_nul_atom():
    # Parser.cpp:87-150: do-while runs once on the terminator -> literal NUL atom, never matched by input
    r = OracleRegex("")
    assert r.states_n == 2
    assert not r.accepts("") and not r.accepts("a")


def test_bytes_outside_domain_reject():
    r = OracleRegex(".*")
    assert r.accepts(b"abc")
    assert not r.accepts(b"a\x00c")
    assert not r.accepts(b"a\x80c")
    assert not r.accepts(b"\xff")


def test_match_lines_splitting():
    import numpy as np
    r = OracleRegex("a*")
    assert list(r.match_lines(b"")) == []
    assert list(r.match_lines(b"\n")) == [1]
    assert list(r.match_lines(b"aa\nb\n\naaa")) == [1, 0, 1, 1]
    assert list(r.match_lines(b"aa\nb\n\naaa\n")) == [1, 0, 1, 1]
    assert list(r.match_lines(np.frombuffer(b"b", dtype=np.uint8))) == [0]


def test_corpus_golden_vectors_reproduce():
    """tests/golden/corpus_golden.json: the generator is frozen (corpus checksum) and the oracle's accept vectors
    over the BASELINE-config corpora reproduce."""
    import json
    import os
    import synth
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "corpus_golden.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        data = synth.corpus(c["kind"], c["seed"], c["bytes"], chunk=c["chunk"], threads=2)
        assert "%016x" % synth.fnv1a(data) == c["corpus_fnv1a"], c["name"]
        if c["bytes"] > (1 << 20):
            continue                                    # the big ones are re-run by the GPU suite
        acc = OracleRegex(c["pattern"]).match_lines(data)
        assert (len(acc), int(acc.sum()), "%016x" % synth.fnv1a(acc)) == (c["lines"], c["accepted"], c["accept_fnv1a"]), c["name"]


def test_vector_or_step_equals_scalar_step(kat):
    """The oracle's BitSet<2> / BitSet<4> step with the reference's vector ORs (BitSet.cc:8-21: _mm_or_si128, _mm256_or_si256 -
    what bench.py's cpu_baseline runs where the CPU has AVX2) and with scalar words: the same answers on the known answers of those
    two classes and on random text."""
    import random
    import pyoracle
    rng = random.Random(5)
    pats = [k["pattern"] for k in kat["kat"] if 64 < OracleRegex(k["pattern"]).states_n <= 256]
    assert len(pats) >= 4
    was = pyoracle.simd()
    try:
        for p in pats:
            o = OracleRegex(p)
            assert o.set_class in (2, 4)
            texts = [k2 for k in kat["kat"] if k["pattern"] == p for k2 in k.get("accepts", []) + k.get("rejects", [])]
            alphabet = "".join(sorted(set("".join(texts)) | set("ab:/.x"))) or "ab"
            texts += ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, 90))) for _ in range(300)]
            got = []
            for on in (False, True):
                pyoracle.set_simd(on)
                got.append([o.accepts(t) for t in texts])
            assert got[0] == got[1], p
    finally:
        pyoracle.set_simd(was)


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    """Sanitizers run on the CPU build only: compile the oracle with -fsanitize=address,undefined and compile a
    batch of patterns (incl. the 7786-state keyword set and every error path) + match a few strings."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = tmp_path / "drv.c"
    drv.write_text(r'''
#include "rr_oracle.h"
#include <stdio.h>
#include <string.h>
int main(void) {
    char line[70000], err[200];
    while (fgets(line, sizeof line, stdin)) {
        line[strcspn(line, "\n")] = 0;
        rro_nfa *n = rro_compile(line, err, sizeof err);
        if (!n) { printf("E\n"); continue; }
        const char *t[] = {"", "a", "abc", "aaaa", "k17", "http://a.bc", "x@y.z"};
        int m = 0;
        for (unsigned i = 0; i < sizeof t / sizeof *t; i++) m += rro_accepts(n, (const uint8_t *)t[i], strlen(t[i]));
        printf("%u %d\n", rro_states_n(n), m);
        rro_free(n);
    }
    return 0;
}''')
    exe = tmp_path / "drv"
    subprocess.check_call(["gcc", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(root, "oracle"),
                           "-o", str(exe), str(drv), os.path.join(root, "oracle", "rr_oracle.c")])
    import json
    with open(os.path.join(root, "tests", "golden", "kat.json")) as f:
        kat = json.load(f)
    pats = [k["pattern"] for k in kat["kat"] if "\n" not in k["pattern"]] + [k["pattern"] for k in kat["big_states"]]
    pats += ["a)", "|a", "a|", "()", "*a", "[", "a{2", "(|a)", "a(|b)", "[0-9][b-d](0+c*1[^a]|[0-9]{1,}([a-c]{1,6})?|k*[^a][0-9]*)?[b-d]?"]
    p = subprocess.run([str(exe)], input=("\n".join(pats) + "\n").encode("latin-1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert len(p.stdout.decode().strip().split("\n")) == len(pats)
