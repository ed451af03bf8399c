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
