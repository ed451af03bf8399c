"""Second, independent oracle: Python's `re.fullmatch` against the C oracle.

The survey found `re.fullmatch` to reproduce the reference's accept vector bit for bit on two 16 MiB corpora; for
automata beyond 256 reference states (configs C4/C5: the reference itself is broken there) it is the only check that
does not share code or author with oracle/rr_oracle.c.  `re` stays in this container: nothing here runs on the GPU box.
The reference dialect is translated, not passed through: `\\x` is "literal x" (Parser.cpp:88-91), `.` and `[^...]` range
over all 128 codes incl. newline (Parser.cpp:106-109), `{m,n}` with n <= m means {m} (Parser.cpp:133).  Patterns using
the reference's quirks (anchors that never match, `{0,n}`, `[\\]]`) are left to the known answers of kat.json.
"""
import json
import os
import random
import re

import numpy as np
import pytest

import synth
from patterns import KAT, random_pattern, random_text
from pyoracle import OracleError, OracleRegex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def translate(p):
    """Reference dialect -> Python `re` source, or None where the reference's behaviour is a quirk."""
    out, i, n = [], 0, len(p)
    while i < n:
        c = p[i]
        if c == "\\":
            if i + 1 >= n:
                return None
            out.append(re.escape(p[i + 1]))
            i += 2
        elif c == "[":
            j = p.find("]", i + 1)
            if j < 0 or "\\" in p[i:j + 1]:
                return None
            body = p[i + 1:j]
            neg = body.startswith("^")
            if neg:
                body = body[1:]
            if not body:
                return None
            k, cls = 0, []
            while k < len(body):
                if k + 2 < len(body) and body[k + 1] == "-":
                    cls.append(re.escape(body[k]) + "-" + re.escape(body[k + 2]))
                    k += 3
                else:
                    cls.append(re.escape(body[k]))
                    k += 1
            out.append("[" + ("^" if neg else "") + "".join(cls) + "]")
            i = j + 1
        elif c == "{":
            j = p.find("}", i)
            if j < 0:
                return None
            body = p[i + 1:j]
            if "," in body:
                lo, hi = body.split(",", 1)
                if not lo.isdigit() or int(lo) < 1:
                    return None
                if hi == "":
                    out.append("{%d,}" % int(lo))
                elif not hi.isdigit():
                    return None
                elif int(hi) <= int(lo):
                    out.append("{%d}" % int(lo))
                else:
                    out.append("{%d,%d}" % (int(lo), int(hi)))
            else:
                if not body.isdigit() or int(body) < 1:
                    return None
                out.append("{%d}" % int(body))
            i = j + 1
        elif c in "^$":
            return None
        elif c in "()|*+?.":
            out.append(c)
            i += 1
        else:
            out.append(re.escape(c))
            i += 1
    return "".join(out)


def reenters_initial(p):
    """True if some operand (the pattern, a group, an alternative) BEGINS with a starred atom.  The reference's Kleene
    star adds back edges into the operand's own initial state (NFA.cc:150-157 via skip<false>, NFA.cc:108-121) instead
    of a fresh one, and `?`, `|` and `{m,n}` then make that re-entered state final or give it the other operand's
    transitions (NFA.cc:138-149): `(b*a)?` accepts "b" and `b*a|d` accepts "bd" in the reference (and in the oracle and
    the product, which reproduce it table for table).  Such patterns have no `re` equivalent and are skipped here."""
    def atom_end(i):
        if i >= len(p):
            return i
        c = p[i]
        if c == "\\":
            return i + 2
        if c == "[":
            j = p.find("]", i + 1)
            return len(p) if j < 0 else j + 1
        if c == "(":
            depth, j = 1, i + 1
            while j < len(p) and depth:
                if p[j] == "\\":
                    j += 1
                elif p[j] == "[":
                    k = p.find("]", j + 1)
                    j = len(p) if k < 0 else k
                elif p[j] == "(":
                    depth += 1
                elif p[j] == ")":
                    depth -= 1
                j += 1
            return j
        return i + 1
    starts, i = [0], 0
    while i < len(p):
        c = p[i]
        if c == "\\":
            i += 2
            continue
        if c == "[":
            j = p.find("]", i + 1)
            i = len(p) if j < 0 else j + 1
            continue
        if c in "(|":
            starts.append(i + 1)
        i += 1
    for s0 in starts:
        e = atom_end(s0)
        if e < len(p) and (p[e] == "*" or re.match(r"\{\d+,\}", p[e:])):
            return True
    return False


def py_accept_lines(pattern, data):
    rx = re.compile(translate(pattern).encode("latin-1"), re.DOTALL)
    lines = bytes(data).split(b"\n")
    if lines and lines[-1] == b"" and len(data) and data[-1] == 10:
        lines.pop()
    # (the synthetic corpora are 7-bit and hold no NUL: the out-of-domain rule does not come into play)
    return np.fromiter((1 if rx.fullmatch(t) is not None else 0 for t in lines), dtype=np.uint8, count=len(lines))


def test_translate_handles_the_config_patterns():
    with open(os.path.join(ROOT, "tests", "golden", "corpus_golden.json")) as f:
        for c in json.load(f)["cases"]:
            assert translate(c["pattern"]) is not None, c["name"]


def test_all_six_golden_corpora_oracle_equals_python_re():
    """Every BASELINE config's golden corpus, incl. a{1,300} (899 reference states) and both keyword sets (7786 / 7790):
    C oracle == re.fullmatch, line by line, and both equal the committed golden count."""
    with open(os.path.join(ROOT, "tests", "golden", "corpus_golden.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        data = synth.corpus(c["kind"], c["seed"], c["bytes"], chunk=c["chunk"], threads=2)
        want = py_accept_lines(c["pattern"], data)
        got = OracleRegex(c["pattern"]).match_lines(data)
        assert got.shape == want.shape, c["name"]
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (c["name"], "line", int(bad[0]))
        assert int(got.sum()) == c["accepted"], c["name"]


def test_larger_samples_of_the_roaring_class_configs():
    """The >256-state classes on more data than the golden fixtures hold (the reference cannot run these at all)."""
    from patterns import K1000, K1000_CONTAINS
    for kind, pattern, nbytes, seed in (("arepeat", "a{1,300}", 1 << 20, 31), ("kwlines", K1000, 256 << 10, 32), ("kwlog", K1000_CONTAINS, 512 << 10, 33)):
        data = synth.corpus(kind, seed, nbytes, threads=2)
        want = py_accept_lines(pattern, data)
        got = OracleRegex(pattern).match_lines(data)
        assert (got == want).all(), kind
        assert 0 < int(got.sum()) < len(got), kind


def test_kat_answers_agree_with_python_re_where_the_dialect_translates():
    n = 0
    for k in KAT["kat"]:
        t = translate(k["pattern"])
        if t is None:
            continue
        rx = re.compile(t, re.DOTALL)
        for s in k["accepts"]:
            assert rx.fullmatch(s) is not None, (k["pattern"], s)
        for s in k["rejects"]:
            assert rx.fullmatch(s) is None, (k["pattern"], s)
        n += 1
    assert n >= 35


def test_the_star_quirk_is_what_the_filter_says_it_is():
    assert reenters_initial("(b*a)?") and reenters_initial("b*a|d") and reenters_initial("x((ab)*c|d)")
    assert not reenters_initial("(ab*)?") and not reenters_initial("a(b|cd*)")
    assert OracleRegex("(b*a)?").accepts("b") and OracleRegex("b*a|d").accepts("bd")       # the reference's semantics
    assert not OracleRegex("(ba)?").accepts("b") and not OracleRegex("ba|d").accepts("bd")


def test_random_patterns_oracle_equals_python_re():
    rng = random.Random(4242)
    done = 0
    while done < 150:
        p = random_pattern(rng)
        t = translate(p)
        if t is None or reenters_initial(p):
            continue
        try:
            o = OracleRegex(p)
        except OracleError:
            continue
        if o.states_n > 600:
            continue
        rx = re.compile(t, re.DOTALL)
        for _ in range(60):
            s = random_text(rng, "abcxk01.d", rng.choice([2, 6, 12, 30]))
            assert o.accepts(s) == (rx.fullmatch(s) is not None), (p, s)
        done += 1
