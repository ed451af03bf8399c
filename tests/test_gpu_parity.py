"""GPU parity: the HIP path (through the C ABI) against the oracle, bit-exact on the accept vectors.
Every case runs BOTH device engines (shift-and NFA and table DFA) where the automaton admits them."""
import random

import numpy as np
import pytest

import roaringregex_amd as rr
from patterns import EMAIL, K1000, K1000_CONTAINS, KAT, U2, random_pattern, random_text
from pyoracle import OracleError, OracleRegex

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROUND, LANES = 128, 1024                         # kernel geometry (device.hpp): bytes per round, lanes per workgroup
STRIPES = (1024, 4096, 16384)                    # bytes per lane; every corpus test runs at each of them


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"


def engines_for(pattern, small=False):
    """One compiled regex per device engine that admits the automaton.  The group- and wave-resident NFAs are the
    engines of big automata (a whole workgroup per string in the block form), so they only join on small inputs."""
    out = []
    kinds = [rr.ENGINE_NFA, rr.ENGINE_DFA, rr.ENGINE_DFA2, rr.ENGINE_DFA_GLOBAL] + ([rr.ENGINE_NFA_WAVE, rr.ENGINE_NFA_BLOCK, rr.ENGINE_NFA_SPARSE] if small else [])
    for e in kinds:
        try:
            out.append(rr.RRegex(pattern, e))
        except rr.RRegexError as err:
            assert "too large" in str(err)
    assert out, pattern
    return out


def check(pattern, data, oracle=None, stripes=(0,)):
    data = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else data
    o = oracle or OracleRegex(pattern)
    want = o.match_lines(data)
    dev = torch.from_numpy(np.array(data, copy=True)).cuda() if len(data) else torch.empty(0, dtype=torch.uint8, device="cuda")
    engines = engines_for(pattern, small=len(data) <= (1 << 18))
    for stripe in stripes:
        corpus = rr.Corpus(dev, stripe=stripe)
        assert corpus.num_lines == len(want)
        for r in engines:
            got = r.match_corpus(corpus).cpu().numpy()
            assert got.shape == want.shape, (pattern[:40], r.engine_name)
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, (pattern[:40], r.engine_name, "stripe", corpus.stripe, "first bad line", int(bad[0]), "of", len(want))
    return want


# ------------------------------------------------------------------------------------------ known answers
def test_kat_batch():
    for k in KAT["kat"]:
        texts = [t for t in k["accepts"] + k["rejects"] if "\n" not in t]
        data = ("\n".join(texts) + "\n").encode("latin-1")
        want = check(k["pattern"], data)
        exp = [1] * len([t for t in k["accepts"] if "\n" not in t]) + [0] * len([t for t in k["rejects"] if "\n" not in t])
        assert list(want) == exp, k["pattern"]


def test_kat_iterator_facade():
    # regex.h:113-122,150-165: it = r.get_acceptance_iter(text); it++; *it
    for k in KAT["kat"][:30] + KAT["kat"][-4:]:
        for r in engines_for(k["pattern"], small=True):
            for t, want in [(t, True) for t in k["accepts"]] + [(t, False) for t in k["rejects"]]:
                it = r.get_acceptance_iter(t)
                it.advance()
                m = it.value()
                assert (m is not None) == want, (k["pattern"], t, r.engine_name)
                if m is not None:
                    assert m.start == 0 and m.end == len(t) and m.str() == t.encode("latin-1")
                it.advance()                                        # idempotent after the terminator
                assert (it.value() is not None) == want


def test_iterator_before_advance_reflects_initial_set():
    assert rr.RRegex("a*").get_acceptance_iter("b").value() is not None     # nullable: accepts before ++
    assert rr.RRegex("a").get_acceptance_iter("a").value() is None
    it = rr.RRegex("ab").get_acceptance_iter("ab")
    c = it.create_copy()
    it.advance()
    assert it.value() is not None and c.value() is None                     # copies are independent


def test_broken_reference_class_uses_intended_semantics():
    for k in KAT["broken_reference"]:
        for r in engines_for(k["pattern"]):
            it = r.get_acceptance_iter(k["text"]).advance()
            assert (it.value() is not None) == k["intended"]


# ------------------------------------------------------------------------------------------ edge cases
def test_empty_and_degenerate_inputs():
    for p in ["a*", "a", ".*"]:
        check(p, b"")
        check(p, b"\n")
        check(p, b"\n\n\n")
        check(p, b"a")
        check(p, b"a\n")
        check(p, b"\na")
        check(p, b"b\n\na\n\n")
    check("a*", b"\n" * 5000)                       # only empty lines: one per byte
    check("a*", b"a\n" * 40000)                     # 2-byte lines across several tiles


def test_bytes_outside_the_domain_reject_their_line():
    data = b"abc\nab\x00c\nabc\nab\x80c\n\xffabc\nabc"
    want = check(".*", data)
    assert list(want) == [1, 0, 1, 0, 0, 1]


def _boundary_corpus(rng, total, cut_positions, alphabet=b"ab"):
    a = np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), size=total)].copy()
    cuts = np.fromiter((p for p in cut_positions if 0 <= p < total), dtype=np.int64)
    a[cuts] = 10
    return a


def test_lines_cut_exactly_at_round_stripe_and_workgroup_boundaries():
    rng = np.random.default_rng(5)
    for stripe in STRIPES:
        total = 5 * stripe + 777
        for delta in (-2, -1, 0, 1):
            cuts = [k * ROUND + delta for k in range(1, total // ROUND + 1, 3)] + [k * stripe + delta for k in range(1, 6)]
            check("(a|b)*abb(a|b)*", _boundary_corpus(rng, total, cuts), stripes=(stripe,))
        # newlines ONLY at stripe boundaries (+delta): every lane starts mid-line or exactly on a line start
        for delta in (-1, 0, 1):
            data = _boundary_corpus(rng, 9 * stripe + 5, [k * stripe + delta for k in range(1, 10)])
            check("(a|b)*abb(a|b)*", data, stripes=(stripe,))
        # sizes that leave a partial round / partial stripe at the end
        for n in (1, 15, 16, 127, 128, 129, stripe - 1, stripe + 1, stripe + 127, 3 * stripe + 128 + 17):
            check("(a|b)*abb(a|b)*", _boundary_corpus(rng, n, range(7, n, 41)), stripes=(stripe,))
    # corpus sizes that are exact multiples of the stripe / of a whole workgroup, with and without trailing newline
    stripe = 1024
    for n in (stripe, 2 * stripe, LANES * stripe, LANES * stripe + stripe):
        data = _boundary_corpus(rng, n, range(50, n, 97))
        check("(a|b)*abb(a|b)*", data, stripes=(stripe,))
        data[-1] = 10
        check("(a|b)*abb(a|b)*", data, stripes=(stripe,))


def test_long_lines_across_many_stripes():
    rng = np.random.default_rng(6)
    for stripe in STRIPES:
        parts = []
        for n in (ROUND - 1, ROUND, ROUND + 1, 5 * ROUND, stripe - 1, stripe, stripe + 1, stripe + ROUND + 5, 3 * stripe + 300,
                  17, 2 * stripe, 40 * stripe + 3, 5):
            parts.append(np.frombuffer(b"ab", dtype=np.uint8)[rng.integers(0, 2, size=n)])
            parts.append(np.array([10], dtype=np.uint8))
        data = np.concatenate(parts)
        check("(a|b)*abb(a|b)*", data, stripes=(stripe,))
        check("(a|b)*", data, stripes=(stripe,))
    # one single unterminated string much longer than a stripe (BASELINE config 1 shape)
    one = np.frombuffer(b"abc", dtype=np.uint8)[rng.integers(0, 3, size=7 * 4096 + 123)]
    check("(a|b|c)*abc", one, stripes=STRIPES)


def test_many_short_lines_overflow_the_result_register():
    # > 16 line ends inside one 16-byte unit / > 32 inside one round: exercises the early flush
    check("a*", b"\n" * (3 * 4096 + 5), stripes=STRIPES)
    check("a?", (b"a\n" * 40000) + b"\n\n\na", stripes=STRIPES)
    rng = np.random.default_rng(8)
    data = np.frombuffer(b"a\n\n", dtype=np.uint8)[rng.integers(0, 3, size=6 * 4096 + 11)]
    check("a{1,3}", data, stripes=STRIPES)


def test_ragged_random_lines_small_alphabet():
    rng = random.Random(7)
    for p in ["ab*c", "(ab|cd)+", "a{2,5}b?", "[ab]+c[ab]*", "(a|b)*abb"]:
        lines = [random_text(rng, "abcd", rng.choice([0, 1, 3, 8, 40, 200])) for _ in range(4000)]
        check(p, ("\n".join(lines)).encode())


# ------------------------------------------------------------------------------------------ random patterns
def test_random_patterns_against_oracle():
    rng = random.Random(99)
    done = 0
    while done < 60:
        p = random_pattern(rng)
        try:
            o = OracleRegex(p)
        except OracleError:
            continue
        if o.states_n > 300:
            continue
        try:
            rr.RRegex(p)
        except rr.RRegexError:
            continue
        lines = [random_text(rng, "abcxk01.d", rng.choice([2, 6, 12, 30])) for _ in range(1500)]
        check(p, ("\n".join(lines) + "\n").encode(), oracle=o)
        done += 1


# ------------------------------------------------------------------------------------------ BASELINE configs
def test_config_corpora_small():
    import synth
    cases = [("email", EMAIL, 4 << 20), ("url", U2, 4 << 20), ("arepeat", "a{1,300}", 96 << 10),
             ("kwlines", K1000, 64 << 10), ("kwlog", K1000_CONTAINS, 128 << 10), ("c1", "abc", 1 << 20)]
    for kind, pattern, nbytes in cases:
        data = synth.corpus(kind, 11, nbytes)
        want = check(pattern, data, stripes=(0, 4096) if nbytes > (1 << 20) else STRIPES)
        if kind in ("email", "url"):
            assert 0.4 < want.mean() < 0.6


def test_extents_api_treats_newline_as_an_ordinary_byte():
    items = [b"a\nb", b"ab", b"", b"\n", b"a\n\nb", b"x"]
    blob = b"".join(items)
    off = np.cumsum([0] + [len(i) for i in items]).astype(np.int64)
    data = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).cuda()
    offs = torch.from_numpy(off).cuda()
    o = OracleRegex("a.*b")
    want = [int(o.accepts(i)) for i in items]
    for r in engines_for("a.*b"):
        got = r.match_extents(data, offs).cpu().numpy()
        assert list(got) == want, r.engine_name


def test_match_host_convenience():
    data = b"abc\nabd\nabc"
    assert list(rr.RRegex("abc").match_host(data)) == [1, 0, 1]
    assert list(rr.RRegex("abc").match_host(b"")) == []


# ------------------------------------------------------------------------------------------ full-size properties
def test_full_size_email_config_properties():
    """BASELINE config 2 at its full 1 GiB: the oracle cannot finish that in seconds, so check size-independent
    properties: (1) both engines agree bit for bit; (2) the corpus is a concatenation of independently generated
    1 MiB chunks, so the accept vector of chunk j taken alone must reappear at chunk j's line offset; a sample
    of chunks is checked against the ORACLE; (3) the accept fraction stays near one half."""
    import synth
    n = 1 << 30
    host = synth.corpus("email", 1, n)
    dev = torch.from_numpy(host).cuda()
    corpus = rr.Corpus(dev)
    nfa = rr.RRegex(EMAIL, rr.ENGINE_NFA).match_corpus(corpus)
    dfa = rr.RRegex(EMAIL, rr.ENGINE_DFA).match_corpus(corpus)
    assert torch.equal(nfa, dfa)
    acc = dfa.cpu().numpy()
    assert 0.45 < acc.mean() < 0.55
    chunk = 1 << 20
    nl_before = np.cumsum(np.add.reduceat((host == 10).astype(np.int64), np.arange(0, n, chunk)))
    o = OracleRegex(EMAIL)
    for j in (0, 1, 17, 511, 1023):
        piece = host[j * chunk:(j + 1) * chunk]
        want = o.match_lines(piece)                 # every chunk ends with '\n' by construction
        first = 0 if j == 0 else int(nl_before[j - 1])
        assert (acc[first:first + len(want)] == want).all(), j


def test_corpus_golden_vectors_on_the_gpu():
    """The committed golden accept vectors (tests/golden/corpus_golden.json, generated by the oracle) through the
    HIP path, both engines, without running the oracle here."""
    import json
    import os
    import synth
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "corpus_golden.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        data = synth.corpus(c["kind"], c["seed"], c["bytes"], chunk=c["chunk"], threads=2)
        assert "%016x" % synth.fnv1a(data) == c["corpus_fnv1a"]
        corpus = rr.Corpus(torch.from_numpy(data).cuda())
        assert corpus.num_lines == c["lines"]
        for r in engines_for(c["pattern"]):
            acc = r.match_corpus(corpus).cpu().numpy()
            assert (int(acc.sum()), "%016x" % synth.fnv1a(acc)) == (c["accepted"], c["accept_fnv1a"]), (c["name"], r.engine_name)


def test_full_size_url_config_properties():
    """BASELINE config 3 (the headline) at its full 8 GiB: engines agree bit for bit on the bitmap; sampled 1 MiB
    chunks (each ends with a newline by construction) reproduce the ORACLE's vector at their line offsets."""
    import synth
    n = 8 << 30
    host = synth.corpus("url", 2, n)
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    step = 1 << 30
    for off in range(0, n, step):
        dev[off:off + step].copy_(torch.from_numpy(host[off:off + step]))
    corpus = rr.Corpus(dev)
    dfa_bits = rr.RRegex(U2, rr.ENGINE_DFA).match_corpus_bits(corpus)
    nfa_bits = rr.RRegex(U2, rr.ENGINE_NFA).match_corpus_bits(corpus)
    assert torch.equal(dfa_bits, nfa_bits)
    acc = rr.RRegex(U2).match_corpus(corpus)
    assert 0.45 < float(acc.float().mean()) < 0.55
    chunk = 1 << 20
    o = OracleRegex(U2)
    for j in (0, 1, 4095, 8191):
        piece = host[j * chunk:(j + 1) * chunk]
        first = int((torch.from_numpy(host[:j * chunk]) == 10).sum()) if j else 0
        want = o.match_lines(piece)
        got = acc[first:first + len(want)].cpu().numpy()
        assert (got == want).all(), j


def _newlines_before(dev, off, piece=1 << 28):
    """number of '\n' in dev[:off], counted on the device piece by piece"""
    total = 0
    for a in range(0, off, piece):
        total += int((dev[a:min(a + piece, off)] == 10).sum(dtype=torch.int64).item())
    return total


def _full_size_properties(kind, seed, pattern, n, engines, chunks, accept_band):
    """Shared body of the full-size tests of the >256-state configs: every engine gives the SAME bitmap; sampled 1 MiB
    chunks (each ends with a newline by construction) reproduce the ORACLE's vector at their line offsets."""
    import synth
    host = synth.corpus(kind, seed, n)
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    step = 1 << 30
    for off in range(0, n, step):
        dev[off:off + step].copy_(torch.from_numpy(host[off:off + step]))
    corpus = rr.Corpus(dev)
    ref_bits = None
    for e in engines:
        r = rr.RRegex(pattern, e)
        bits = r.match_corpus_bits(corpus).clone()
        if ref_bits is None:
            ref_bits, ref_name = bits, r.engine_name
        else:
            assert torch.equal(bits, ref_bits), (kind, r.engine_name, "vs", ref_name)
        del bits
    nlines = corpus.num_lines
    acc = rr.RRegex(pattern).match_corpus(corpus)                     # byte-per-line form of the same result
    ones = int(acc.sum(dtype=torch.int64).item())
    del acc
    assert accept_band[0] < ones / nlines < accept_band[1], (kind, ones, nlines)
    chunk = 1 << 20
    o = OracleRegex(pattern)
    for j in chunks:
        piece = host[j * chunk:(j + 1) * chunk]
        first = _newlines_before(dev, j * chunk)
        want = o.match_lines(piece)
        w0, w1 = first >> 5, (first + len(want) + 31) >> 5
        words = ref_bits[w0:w1].cpu().numpy().view(np.uint32)
        got = ((words[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.uint8).reshape(-1)[first - 32 * w0:][:len(want)]
        assert (got == want).all(), (kind, "chunk", j)


def test_full_size_arepeat_config_properties():
    """BASELINE config 4 (a{1,300}, 899 reference states: the Roaring class) at its per-GPU size of 1 GiB."""
    _full_size_properties("arepeat", 3, "a{1,300}", 1 << 30, (rr.ENGINE_DFA2, rr.ENGINE_DFA, rr.ENGINE_NFA), (0, 1, 513, 1023), (0.6, 0.9))


def test_full_size_kwlines_config_properties():
    """BASELINE config 5(i) (k1|...|k1000, 7786 reference states) at its per-GPU size of 8 GiB: 1.58 G lines of 5.4
    bytes, i.e. nearly every result word leaves the workgroup's LDS window (the direct-to-memory overflow path)."""
    _full_size_properties("kwlines", 4, K1000, 8 << 30, (rr.ENGINE_DFA2, rr.ENGINE_DFA, rr.ENGINE_NFA), (0, 2, 4097, 8191), (0.4, 0.6))


def test_full_size_kwlog_config_properties():
    """BASELINE config 5(ii) (.*(k1|...|k1000).* over log lines) at 8 GiB per GPU."""
    _full_size_properties("kwlog", 4, K1000_CONTAINS, 8 << 30, (rr.ENGINE_DFA2, rr.ENGINE_DFA, rr.ENGINE_NFA), (0, 3, 4098, 8191), (0.2, 0.8))


def test_long_line_variants_of_the_email_and_url_configs():
    """"also report a long-line variant (>= 256 B/line) for C2/C3" - the corpora bench.py times as email_long and
    url_long (lines of 290-600 and 390-740 bytes, half of them corrupted), 1 GiB each: every engine the same bitmap, the oracle on
    sampled chunks, and the first-match search on a 64 MiB slice against the accept bits (an accepted line's first match... need
    not be the whole line, but a line has a match iff some substring is accepted, and the whole line is one)."""
    for kind, pattern in (("email_long", EMAIL), ("url_long", U2)):
        _full_size_properties(kind, 2, pattern, 1 << 30, (rr.ENGINE_DFA2, rr.ENGINE_DFA, rr.ENGINE_NFA), (0, 511, 1023), (0.4, 0.6))
        import synth
        host = synth.corpus(kind, 2, 64 << 20)
        dev = torch.from_numpy(host).cuda()
        corpus = rr.Corpus(dev)
        r = rr.RRegex(pattern)
        acc = r.match_corpus(corpus)
        st, en = r.search_corpus(corpus)
        assert bool(((st >= 0) | (acc == 0)).all()), kind                  # an accepted line holds a match
        o = OracleRegex(pattern)
        lines = host[: 1 << 16].tobytes().split(b"\n")[:24]               # (the oracle's brute force is quadratic in the match end)
        s_, e_ = st[:len(lines)].cpu().tolist(), en[:len(lines)].cpu().tolist()
        for ln, a, b in zip(lines, s_, e_):
            if a >= 0:
                assert o.accepts(ln[a:b]) and not any(o.accepts(ln[a:k]) for k in range(a, b)), (kind, ln, a, b)
                assert not any(o.accepts(ln[j:k]) for k in range(0, b) for j in range(0, k + 1)) , (kind, ln, a, b)   # no earlier end
                assert not any(o.accepts(ln[j:b]) for j in range(0, a)), (kind, ln, a, b)                              # no earlier start
            else:
                assert not any(o.accepts(ln[j:k]) for k in range(0, min(len(ln), 40) + 1) for j in range(0, k + 1)), (kind, ln)


def test_small_corpora_take_short_stripes():
    """The automatic stripe: 512 bytes while the corpus leaves no more than 2^18 lanes (a lane steps its stripe as one chain of dependent
    lookups: short stripes are what makes a small corpus fast), 2 KiB at 1 GiB as before; the stripe never changes a result."""
    import synth
    for nbytes, want_stripe in ((64 << 10, 512), (16 << 20, 512), (128 << 20, 512), (256 << 20, 1024)):
        host = synth.corpus("url", 13, nbytes)
        dev = torch.from_numpy(host).cuda()
        auto = rr.Corpus(dev)
        assert auto.stripe == want_stripe, (nbytes, auto.stripe)
        r = rr.RRegex(U2)
        got = r.match_corpus_bits(auto)
        assert torch.equal(got, r.match_corpus_bits(rr.Corpus(dev, stripe=4096)))
        head = host[:1 << 18]
        cut = int(np.flatnonzero(head == 10)[-1]) + 1
        want = OracleRegex(U2).match_lines(head[:cut])
        assert (_bits_to_bytes(got, len(want)) == want).all()
        del dev, auto


def test_indexed_match_replays_from_a_hip_graph():
    """rrx_match_corpus is one memset and one kernel launch on the caller's stream - no host synchronisation, no allocation once the
    tables are up - so a scan loop over several patterns can be captured in a hipGraph (torch.cuda.CUDAGraph) and replayed: the form
    for launch-bound batches of small corpora.  Three engines (stride-2 table, byte-stride table, NFA lane engine) in one graph; the
    replay must give what the eager calls gave, on every replay."""
    import synth
    dev = torch.from_numpy(synth.corpus("url", 7, 8 << 20)).cuda()
    corpus = rr.Corpus(dev)
    regs = [rr.RRegex(U2), rr.RRegex(EMAIL, rr.ENGINE_DFA), rr.RRegex("(a|b)*a(a|b){40}"), rr.RRegex(".*(k1|k2|k17|k100).*", rr.ENGINE_NFA)]
    for r in regs:
        r.set_background_order(False)                                  # (no thread of the library's own while the capture runs)
        r.set_sampled_table(False)
    nw = regs[0].match_corpus_bits(corpus).numel()
    outs = [torch.zeros(nw, dtype=torch.int32, device="cuda") for _ in regs]
    for r, o in zip(regs, outs):
        r.match_corpus_bits(corpus, out=o)                             # eager: tables uploaded, results to compare with
    torch.cuda.synchronize()
    want = [o.clone() for o in outs]
    assert int(want[0].ne(0).sum()) > 0
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    # (thread_local: a background thread of some other regex of this process allocating meanwhile must not fail the capture)
    with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
        for r, o in zip(regs, outs):
            r.match_corpus_bits(corpus, out=o)
    for _ in range(3):
        for o in outs:
            o.fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        for r, o, w in zip(regs, outs, want):
            assert torch.equal(o, w), r.engine_name


def test_profiled_table_order_changes_no_result():
    """At its first match against a corpus of 64 MiB or more a background thread orders the stride-2 table by a sample of the
    text (rows and columns permuted so that fewer lookups of a half-wave share an LDS bank) and swaps the tables in when it is
    done.  Matches before, during and after the swap must give the same bits: bit-identical to the byte-stride table engine
    (which has no such order) and to the oracle on sampled chunks; the sample's conflict figure must not get worse; a regex
    whose table is small enough to be replicated is left alone."""
    import time
    import synth
    n = 96 << 20
    host = synth.corpus("url", 21, n)
    dev = torch.from_numpy(host).cuda()
    corpus = rr.Corpus(dev)
    ref = rr.RRegex(U2, rr.ENGINE_DFA).match_corpus_bits(corpus).clone()
    r = rr.RRegex(U2)
    assert r.engine_name == "dfa-stride2-table" and r.table_order is None
    out = torch.empty_like(ref)
    t0 = time.time()
    launches = 0
    while r.table_order is None and time.time() - t0 < 20:      # launches race the swap: every one must be right
        assert torch.equal(r.match_corpus_bits(corpus, out=out), ref)
        launches += 1
    order = r.table_order
    assert order is not None and order[1] <= order[0] and order[0] > 1.0, (order, launches)
    for _ in range(3):
        assert torch.equal(r.match_corpus_bits(corpus, out=out), ref)
    acc = r.match_corpus(corpus).cpu().numpy()
    o = OracleRegex(U2)
    chunk = 1 << 20
    for j in (0, 47, 95):
        piece = host[j * chunk:(j + 1) * chunk]
        first = int((host[:j * chunk] == 10).sum())
        want = o.match_lines(piece)
        assert (acc[first:first + len(want)] == want).all(), j
    # the one-shot entry shares the tables
    b2, nl = r.match_device_bits(dev)
    assert nl == corpus.num_lines and torch.equal(b2, ref)
    small = rr.RRegex(EMAIL)                                     # five states: replicated copies instead of an order
    small.match_corpus_bits(corpus)
    time.sleep(0.3)
    assert small.table_order is None


def test_units_handed_out_inside_the_workgroup_change_no_result():
    """RRX_OPT_UNITS_PER_WORKGROUP: the stride-2 kernel with its stripes handed out in units of 64 from a counter in LDS (a wave
    takes several) - against the oracle on ragged text at 512-byte to 4 KiB stripes, workgroups that end inside the corpus, units
    beyond the last stripe; and bit for bit against the one-stripe-per-lane kernel on 96 MiB of the URL corpus."""
    import synth
    rng = np.random.default_rng(61)
    p = "(a|b)*abb(a|b)*"
    o = OracleRegex(p)
    alphabet = np.frombuffer(b"ab\n", dtype=np.uint8)
    for n, probs in ((3_000_017, [0.46, 0.46, 0.08]), (1_200_000, [0.4995, 0.4995, 0.001]), (70_001, [0.3, 0.3, 0.4])):
        data = alphabet[rng.choice(3, size=n, p=probs)].copy()
        want = o.match_lines(data)
        dev = torch.from_numpy(data).cuda()
        for stripe in (512, 1024, 4096):
            corpus = rr.Corpus(dev, stripe=stripe)
            for units in (16, 17, 40, 4096):
                r = rr.RRegex(p)
                assert r.engine_name == "dfa-stride2-table"
                r.set_units_per_workgroup(units)
                got = r.match_corpus(corpus).cpu().numpy()
                assert got.shape == want.shape and (got == want).all(), (n, stripe, units, np.nonzero(got != want)[0][:4])
    data = synth.corpus("url", 2, 96 << 20)
    dev = torch.from_numpy(data).cuda()
    plain = rr.RRegex(U2)
    plain.set_background_order(False)
    for stripe in (512, 2048):
        corpus = rr.Corpus(dev, stripe=stripe)
        ref = plain.match_corpus_bits(corpus).clone()
        for units in (32, 64):
            r = rr.RRegex(U2)
            r.set_background_order(False)
            r.set_units_per_workgroup(units)
            assert torch.equal(r.match_corpus_bits(corpus), ref), (stripe, units)
    with pytest.raises(rr.RRegexError):
        plain.set_units_per_workgroup(1 << 20)


def test_sampled_table_for_automata_that_do_not_determinise():
    """U2(x|y)*x(x|y){30} and (U2)|(x|y)*x(x|y){30}: 2^31 state sets on x/y text, so AUTO leaves them on the NFA lane engine; the sets
    URL text reaches are a hundred.  rrx_learn_table interns those into a table with an ESCAPE state; rrx_match_corpus then runs the
    stride-2 kernel with two result bits per line and the NFA engine decides the escaped lines.  Exact against the oracle on text like
    the sample (no line escapes), on text built to escape on most lines (x/y tails the sample never showed), on a mix, at three
    stripe sizes, with empty lines, lines longer than a stripe and no final newline - and equal to the NFA engine alone."""
    import synth
    rng = random.Random(77)
    url = synth.corpus("url", 3, 3 << 20)
    lines = url.tobytes().split(b"\n")[:-1]
    tail = lambda: bytes(rng.choice(b"xy") for _ in range(rng.randint(0, 80)))
    # (what escapes: x/y runs the sample never showed - behind a valid url for the first pattern, on their own for the second)
    hostile = [ln + tail() for ln in lines[:6000]] + [tail() for _ in range(3000)] + [b"", b"x" * 31, b"y" * 31, b"x" * 5000 + b"y" * 30,
                                                                                      lines[0] + b"x" * 9000, lines[2] + b"\x00x"]
    dirty = hostile[:500] + [b"\x80" + lines[1], lines[3] + b"\xc3\xa9x"]         # bytes >= 0x80: the corpus takes the NFA engine
    mixed = lines[:20000]
    for k in range(0, len(mixed), 37):
        mixed[k] = mixed[k] + tail()
    for pattern in (U2 + "(x|y)*x(x|y){30}", "(" + U2 + ")|(x|y)*x(x|y){30}"):
        o = OracleRegex(pattern)
        r = rr.RRegex(pattern)
        assert r.engine_name == "nfa-shift-and" and r.sampled_table is None
        plain = rr.RRegex(pattern)
        plain.set_sampled_table(False)
        states, open_tr = r.learn_table(url[:1 << 16])
        assert 20 < states < 2000 and open_tr > 0 and r.sampled_table == (states, open_tr)
        # (the hostile text last: a launch that sees more than 5 % of the lines escape RETIRES the table - the launches after it run on
        # the NFA engine, still exact)
        for name, ls, tailnl in (("like the sample", lines[:30000], b"\n"), ("dirty", dirty, b"\n"), ("mixed", mixed, b"\n"), ("hostile", hostile, b"")):
            data = np.frombuffer(b"\n".join(ls) + tailnl, dtype=np.uint8)
            clean = not (data >= 0x80).any()
            want = o.match_lines(data)
            dev = torch.from_numpy(data.copy()).cuda()
            for stripe in (512, 4096, 0):
                corpus = rr.Corpus(dev, stripe=stripe)
                got = r.match_corpus(corpus).cpu().numpy()
                assert got.shape == want.shape and (got == want).all(), (pattern[-24:], name, stripe, np.nonzero(got != want)[0][:5])
                assert torch.equal(plain.match_corpus_bits(corpus), r.match_corpus_bits(corpus)), (name, stripe, clean)
                if clean and name == "hostile":
                    assert r.sampled_escapes() > 1000, "the hostile text is meant to escape"
                if name == "like the sample":
                    assert r.sampled_escapes() < len(ls) // 50
                if name != "hostile":
                    assert not r.sampled_table_retired, name
            # the one-shot entry of a regex on its sampled table: index pass + table kernel (not the NFA engine's single pass)
            bits1, n1 = r.match_device_bits(dev)
            assert n1 == len(want) and (_bits_to_bytes(bits1, n1) == want).all(), (pattern[-24:], name, "one-shot")
        torch.cuda.synchronize()
        r.match_corpus_bits(corpus)                                 # (the launch that looks at the hostile launches' count)
        assert r.sampled_table_retired and r.sampled_table is None
        assert plain.sampled_table is None
    # an automaton that determinises has no use for it; neither has a forced engine
    with pytest.raises(rr.RRegexError):
        rr.RRegex(U2).learn_table(url[:4096])
    with pytest.raises(rr.RRegexError):
        rr.RRegex(U2 + "(x|y)*x(x|y){30}", rr.ENGINE_NFA).learn_table(url[:4096])


def test_sampled_table_learnt_from_the_first_large_corpus():
    """Without rrx_learn_table: the first rrx_match_corpus against a corpus of 64 MiB or more starts the build from the corpus' own
    sample in the background; launches race the swap, every one of them exact (the NFA engine before, the table after)."""
    import synth
    import time
    pattern = "(" + U2 + ")|(x|y)*x(x|y){30}"
    data = synth.corpus("url", 5, 96 << 20)
    dev = torch.from_numpy(data).cuda()
    corpus = rr.Corpus(dev)
    plain = rr.RRegex(pattern)
    plain.set_sampled_table(False)
    ref = plain.match_corpus_bits(corpus).clone()
    head = data[:1 << 20]
    cut = int(np.flatnonzero(head == 10)[-1]) + 1
    want = OracleRegex(pattern).match_lines(head[:cut])
    got = plain.match_corpus(corpus).cpu().numpy()
    assert (got[:len(want)] == want).all() and 0 < want.sum() < len(want)
    r = rr.RRegex(pattern)
    t0 = time.time()
    launches = 0
    while r.sampled_table is None and time.time() - t0 < 30:
        assert torch.equal(r.match_corpus_bits(corpus), ref)
        launches += 1
    assert r.sampled_table is not None, launches
    for _ in range(3):
        assert torch.equal(r.match_corpus_bits(corpus), ref)


def test_sampled_table_is_learnt_again_from_the_corpus_that_retired_it():
    """A table learnt from bare URLs (scheme, host, nothing else), then the URL corpus proper - ports, paths, queries the old table has
    never seen: most lines escape, the table is retired.  The first launch against that corpus after the retirement starts a new build
    from the corpus' own sample; launches race the swap and are exact throughout (the NFA engine while the table is out: equal to a regex
    that never had one, and to the oracle on the head of the corpus); afterwards next to no line escapes, and the bare URLs run on the
    new table too."""
    import synth
    import time
    pattern = U2 + "(x|y)*x(x|y){30}"
    url = synth.corpus("url", 9, 72 << 20)                             # (large enough to carry a text sample)
    rng = random.Random(5)
    bare = np.frombuffer(b"".join(rng.choice([b"http", b"https", b"ftp"]) + b"://" + bytes(rng.choice(b"abcdefghij") for _ in range(rng.randint(2, 9))) +
                                  b"." + rng.choice([b"com", b"org", b"de"]) + rng.choice([b"", b"", b"", b" "]) + b"\n" for _ in range(40000)), dtype=np.uint8)
    dev_a, dev_b = torch.from_numpy(bare.copy()).cuda(), torch.from_numpy(url).cuda()
    corpus_a, corpus_b = rr.Corpus(dev_a), rr.Corpus(dev_b)
    plain = rr.RRegex(pattern)
    plain.set_sampled_table(False)
    ref_a, ref_b = plain.match_corpus_bits(corpus_a).clone(), plain.match_corpus_bits(corpus_b).clone()
    head = url[:1 << 19]
    cut = int(np.flatnonzero(head == 10)[-1]) + 1
    o = OracleRegex(pattern)
    want = o.match_lines(head[:cut])
    got = plain.match_corpus(corpus_b).cpu().numpy()
    assert (got[:len(want)] == want).all()
    assert (plain.match_corpus(corpus_a).cpu().numpy() == o.match_lines(bare)).all()
    for background in (True, False):                                   # (False: the new build runs inside the launch that starts it)
        r = rr.RRegex(pattern)
        r.set_background_order(background)
        first = r.learn_table(bare[:1 << 16])
        assert torch.equal(r.match_corpus_bits(corpus_a), ref_a) and r.sampled_escapes() < corpus_a.num_lines // 50
        assert torch.equal(r.match_corpus_bits(corpus_b), ref_b)          # on the old table: ports, paths and queries escape
        assert r.sampled_escapes() > corpus_b.num_lines // 20             # (more than the 5 % that retire a table)
        torch.cuda.synchronize()
        assert torch.equal(r.match_corpus_bits(corpus_b), ref_b)          # the launch that sees that count: retired
        assert r.sampled_table_retired
        t0 = time.time()
        launches = 0
        while r.sampled_table_retired and time.time() - t0 < 60:           # the launches that start the new build and race it
            assert torch.equal(r.match_corpus_bits(corpus_b), ref_b)
            launches += 1
        assert not r.sampled_table_retired and r.sampled_table is not None and r.sampled_table != first, (launches, r.sampled_table, first)
        for _ in range(3):
            assert torch.equal(r.match_corpus_bits(corpus_b), ref_b)
        assert r.sampled_escapes() < corpus_b.num_lines // 50
        assert torch.equal(r.match_corpus_bits(corpus_a), ref_a) and r.sampled_escapes() < corpus_a.num_lines // 50
        torch.cuda.synchronize()
        assert torch.equal(r.match_corpus_bits(corpus_a), ref_a) and not r.sampled_table_retired


def test_sampled_table_is_not_installed_where_the_text_escapes_from_it():
    """(a|b)*a(a|b){40} over random a/b lines: every line reaches a set no sample has shown, a sampled table would send every line
    through the NFA engine a second time.  The build (started by the first launch against a large corpus) must find that on its own
    sample and leave the regex on the NFA engine; results are the oracle's before and after."""
    import synth
    import time
    pattern = "(a|b)*a(a|b){40}"
    data = synth.corpus("ablines", 7, 80 << 20)
    corpus = rr.Corpus(torch.from_numpy(data).cuda())
    head = data[:256 << 10]
    cut = int(np.flatnonzero(head == 10)[-1]) + 1
    want = OracleRegex(pattern).match_lines(head[:cut])
    r = rr.RRegex(pattern)
    assert r.engine == rr.ENGINE_NFA
    first = r.match_corpus(corpus).cpu().numpy()
    assert (first[:len(want)] == want).all() and 0 < want.sum() < len(want)
    t0 = time.time()
    while r.sampled_table_pending and time.time() - t0 < 60:
        time.sleep(0.05)
    assert not r.sampled_table_pending and r.sampled_table is None
    assert (r.match_corpus(corpus).cpu().numpy() == first).all()


def test_table_order_from_a_caller_sample_after_the_tables_are_up():
    """rrx_order_table on a regex whose tables are already on the device (it matched a small corpus first): the stride-2 arrays
    are uploaded again in the new order and swapped in; results before and after are the oracle's."""
    import synth
    host = synth.corpus("url", 23, 8 << 20)
    want = OracleRegex(U2).match_lines(host[:1 << 20])
    dev = torch.from_numpy(host).cuda()
    corpus = rr.Corpus(dev)
    r = rr.RRegex(U2)
    before = r.match_corpus(corpus).cpu().numpy()
    assert r.table_order is None and (before[:len(want)] == want).all()
    sample = np.ascontiguousarray(host[:256 * 4096].reshape(256, 4096)[:, :256])
    b, a = r.order_table(sample, 256, 256)
    assert a < b
    after = r.match_corpus(corpus).cpu().numpy()
    assert (after == before).all()
    bits, nl = r.match_device_bits(dev)
    assert nl == corpus.num_lines and (_bits_to_bytes(bits, nl) == before).all()


def test_explicit_items_on_two_streams_share_the_scratch_in_order():
    """rrx_match_extents is asynchronous; the item index of a large batch lives in a scratch buffer of the regex handle that
    two streams use in turn, ordered on the device by an event.  Two different batches alternating on two streams, many
    times, without any host synchronisation in between: every result must be its own batch's."""
    import synth
    r = rr.RRegex(EMAIL)
    batches = []
    for seed in (31, 32):
        host = synth.corpus("email", seed, 24 << 20)
        dev = torch.from_numpy(host).cuda()
        nl = torch.nonzero(dev == 10).flatten()
        off = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), nl + 1]).contiguous()
        want = r.match_corpus(rr.Corpus(dev))[:off.numel() - 1].clone()
        batches.append((dev, off, want))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[torch.empty(b[1].numel() - 1, dtype=torch.uint8, device="cuda") for _ in range(6)] for b in batches]
    for it in range(6):
        for k in (0, 1):
            dev, off, _ = batches[k]
            r.match_extents(dev, off, trim=1, out=outs[k][it], stream=streams[k])
    torch.cuda.synchronize()
    for k in (0, 1):
        for it in range(6):
            assert torch.equal(outs[k][it].bool(), batches[k][2].bool()), (k, it)


def test_nul_bytes_in_a_seven_bit_corpus_reach_the_stride2_kernel():
    """A corpus with 0x00 bytes but no byte >= 0x80 keeps the stride-2 engine (has_high stays false): its pair table row
    and column 0 must send the line to the dead state.  All table engines and the NFA, every stripe size."""
    rng = np.random.default_rng(33)
    alphabet = np.frombuffer(b"ab\n\x00", dtype=np.uint8)
    for n in (5, 4096 + 3, 300_001):
        data = alphabet[rng.choice(4, size=n, p=[0.45, 0.45, 0.07, 0.03])].copy()
        assert data.max() < 0x80
        assert rr.RRegex("(a|b)*abb(a|b)*", rr.ENGINE_DFA2).engine_name == "dfa-stride2-table"
        want = check("(a|b)*abb(a|b)*", data, stripes=STRIPES)
        check("(a|b)*", data, stripes=STRIPES)
    lines = [b"ab", b"a\x00b", b"\x00", b"", b"ab\x00", b"\x00ab", b"ab"]
    want = check("(a|b)*", b"\n".join(lines) + b"\n")
    assert list(want) == [1, 0, 0, 1, 0, 0, 1]


def test_short_lines_overflow_the_workgroup_result_window():
    """More lines inside one workgroup than its LDS result window holds (131 k - 620 k): several MiB of 1-2 byte lines at
    stripe 1024 send most result words down the direct global-atomic path.  Table engines and the NFA vs the oracle."""
    rng = np.random.default_rng(34)
    data = np.frombuffer(b"a\n\n", dtype=np.uint8)[rng.integers(0, 3, size=6 << 20)].copy()
    check("a?", data, stripes=(1024, 4096))
    data = np.frombuffer(b"ab\n", dtype=np.uint8)[rng.integers(0, 3, size=(3 << 20) + 77)].copy()
    check("(a|b)b?", data, stripes=(1024,))


def test_large_global_table_automaton_through_every_entry_point():
    """A 3000-word alternation: ~9.9 k interned sets, far beyond LDS -> dfa-global-table.  The batch kernel, the extents
    kernel and the reference's own API (get_acceptance_iter(text)++; *it, regex.h:225-227) must all work on it: the
    single-string entries read the plain table from HBM/L2 when it does not fit LDS."""
    rng = random.Random(5)
    words = sorted({"".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(4, 9))) for _ in range(3000)})
    pattern = "|".join(words)
    r = rr.RRegex(pattern)
    assert r.engine_name == "dfa-global-table"
    o = OracleRegex(pattern)
    items = []
    for w in rng.sample(words, 150):
        items += [w, w + "x", w[:-1], w[1:], w + w]
    items += ["", "zzzz"]
    want = [int(o.accepts(t)) for t in items]
    assert 100 < sum(want) < len(want)
    blob = "".join(items).encode()
    off = np.cumsum([0] + [len(t) for t in items]).astype(np.int64)
    got = r.match_extents(torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).cuda(), torch.from_numpy(off).cuda()).cpu().numpy()
    assert list(got) == want
    got = r.match_corpus(rr.Corpus(("\n".join(items)).encode())).cpu().numpy()
    assert list(got) == want
    for t, w in list(zip(items, want))[:40]:
        m = r.get_acceptance_iter(t).advance().value()
        assert (m is not None) == bool(w), t
        if m is not None:
            assert (m.start, m.end) == (0, len(t))
    dev = torch.from_numpy(np.frombuffer(words[7].encode(), dtype=np.uint8).copy()).cuda()
    assert r.match_string(dev) is True


def test_wave_cooperative_engine_on_large_automata():
    """Automata beyond 512 positions whose subset construction explodes (AUTO -> nfa-group-cooperative), against the
    oracle: batch kernel at several stripe sizes, and the single-string entry."""
    rng = random.Random(77)
    p = "(a|b)*a(a|b){600}"
    r = rr.RRegex(p)
    assert r.engine_name == "nfa-group-cooperative"
    o = OracleRegex(p)
    lines = []
    for n in (0, 1, 600, 601, 602, 603, 700, 1300, 5000):
        for _ in range(3):
            lines.append("".join(rng.choice("ab") for _ in range(n)))
        lines.append("a" * n)
        lines.append("b" * n)
    data = ("\n".join(lines)).encode()
    want = o.match_lines(np.frombuffer(data, dtype=np.uint8))
    assert 0 < want.sum() < len(want)
    dev = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    for stripe in STRIPES:
        got = r.match_corpus(rr.Corpus(dev, stripe=stripe)).cpu().numpy()
        assert (got == want).all(), stripe
    for t in lines[:12]:
        assert (r.get_acceptance_iter(t).advance().value() is not None) == o.accepts(t)


def _coop_engine_case(pattern, engine, engine_name, lengths, seed, alphabet="ab", stripes=STRIPES, per_length=2, facade=4,
                      boundary_lines=12, boundary_deltas=(-1, 0, 1)):
    """One cooperative-engine build against the ORACLE through every entry: the batch kernel at several stripe sizes on a
    ragged corpus with and without a final '\n', a corpus whose line ends sit exactly on stripe boundaries (+-1), the
    one-shot entry (two passes for these engines), explicit items, and the iterator facade."""
    rng = random.Random(seed)
    r = rr.RRegex(pattern, engine)
    assert r.engine_name == engine_name, (pattern[:30], r.engine_name)
    o = OracleRegex(pattern)
    lines = []
    for n in lengths:
        for _ in range(per_length):
            lines.append("".join(rng.choice(alphabet) for _ in range(n)))
        lines.append(alphabet[0] * n)
        lines.append(alphabet[-1] * n)
    want_of = {}

    def want(t):
        if t not in want_of:
            want_of[t] = int(o.accepts(t))
        return want_of[t]

    expect = np.array([want(t) for t in lines], dtype=np.uint8)
    assert 0 < expect.sum() < len(expect), (pattern[:30], int(expect.sum()), len(expect))
    for tail in ("", "\n"):
        data = np.frombuffer(("\n".join(lines) + tail).encode("latin-1"), dtype=np.uint8)
        dev = torch.from_numpy(data.copy()).cuda()
        for stripe in stripes:
            corpus = rr.Corpus(dev, stripe=stripe)
            assert corpus.num_lines == len(lines)
            got = r.match_corpus(corpus).cpu().numpy()
            bad = np.nonzero(got != expect)[0]
            assert bad.size == 0, (pattern[:30], engine_name, "stripe", stripe, "tail", repr(tail), "first bad line", int(bad[0]), len(lines[int(bad[0])]))
        bits, nlines = r.match_device_bits(dev)
        assert nlines == len(lines) and (_bits_to_bytes(bits, nlines) == expect).all(), (pattern[:30], "one-shot entry")
    # line ends exactly on / next to stripe boundaries: the same line texts re-cut so that every '\n' sits at k*stripe + d
    for stripe in stripes[:2]:
        for d in boundary_deltas:
            parts, texts, pos = [], [], 0
            for t in lines:
                if not t or len(texts) >= boundary_lines:
                    continue
                end = pos + len(t)                              # where this line's '\n' would fall
                target = ((end + stripe - 1) // stripe) * stripe + d
                pad = target - end
                if pad < 0:
                    pad += stripe
                t2 = t + alphabet[-1] * pad
                texts.append(t2)
                parts.append(t2)
                pos += len(t2) + 1
                if pos > 40 * stripe:
                    break
            data = np.frombuffer(("\n".join(parts) + "\n").encode("latin-1"), dtype=np.uint8)
            exp2 = np.array([want(t) for t in texts], dtype=np.uint8)
            got = r.match_corpus(rr.Corpus(torch.from_numpy(data.copy()).cuda(), stripe=stripe)).cpu().numpy()
            assert (got == exp2).all(), (pattern[:30], engine_name, "boundary cut", stripe, d)
    off = np.cumsum([0] + [len(t) for t in lines]).astype(np.int64)
    blob = np.frombuffer("".join(lines).encode("latin-1"), dtype=np.uint8)
    blob = torch.from_numpy(blob.copy()).cuda() if len(blob) else torch.empty(0, dtype=torch.uint8, device="cuda")
    got = r.match_extents(blob, torch.from_numpy(off).cuda()).cpu().numpy()
    assert (got == expect).all(), (pattern[:30], engine_name, "extents")
    for t in lines[:facade]:
        assert (r.get_acceptance_iter(t).advance().value() is not None) == bool(want(t)), (pattern[:30], "facade", len(t))


def test_group_engine_32_lanes_per_string_and_beyond():
    """match_stripes_group_kernel<G, K, ...>: G lanes x K words per string (r4; K was 2).  16 x 3 words (1025-1536 positions: row_shr:1
    inside a DPP row), 32 x 3 / 32 x 5 / 32 x 8 (two strings per wave, wave_shr:1 carries a word across a DPP row boundary; up to 8192
    positions), with their extents forms; beyond 8192 positions a string takes a whole wave (the wave-resident engine).  These
    patterns are the slot-0 build (only word 0 of a lane carries masks); the general build: the third case and the a{1,n} family.
    Reference: any automaton size, Parser.cpp:165; the loop, NFA.cc:77-85."""
    _coop_engine_case("(a|b)*a(a|b){1500}", rr.ENGINE_AUTO, "nfa-group-cooperative", (0, 1, 1500, 1501, 1502, 1503, 1600, 2100, 3300, 5000), 31)
    assert rr.RRegex("(a|b)*a(a|b){1500}").words_per_set == 47
    _coop_engine_case("(a|b)*a(a|b){3000}", rr.ENGINE_AUTO, "nfa-group-cooperative", (0, 2, 3000, 3001, 3002, 3003, 3100, 4200, 6100, 9000), 32)
    assert rr.RRegex("(a|b)*a(a|b){3000}").words_per_set == 94
    _coop_engine_case("(a|b)*a(a|b){5000}", rr.ENGINE_AUTO, "nfa-group-cooperative", (0, 2, 5000, 5001, 5002, 5003, 5100, 7000, 10100), 35,
                      stripes=(1024, 16384), facade=2, boundary_lines=6)
    _coop_engine_case("(a|b)*a(a|b){8100}", rr.ENGINE_AUTO, "nfa-group-cooperative", (0, 2, 8100, 8101, 8102, 8103, 8200, 16300), 36,
                      stripes=(1024, 16384), facade=2, boundary_lines=4)
    _coop_engine_case("(a|b)*a(a|b){8200}", rr.ENGINE_AUTO, "nfa-wave-resident", (0, 2, 8200, 8201, 8202, 8203, 8300, 16500), 37,
                      stripes=(1024, 16384), per_length=1, facade=2, boundary_lines=4)
    # a text byte outside the pattern's classes in the slot-0 build: the slots without masks cannot be cleared - the line is dead by flag
    _coop_engine_case("(a|b)*a(a|b){1500}", rr.ENGINE_AUTO, "nfa-group-cooperative", (1, 1502, 1503, 1600, 3300), 38, alphabet="abc", stripes=(1024, 16384))
    # exception edges reaching across lanes and rows of the group (not just the add-one chain)
    _coop_engine_case("((a|b)*a(a|b){700}c|(b|c)*b(b|c){800}a)*", rr.ENGINE_NFA_WAVE, "nfa-group-cooperative",
                      (0, 3, 702, 703, 802, 803, 1505, 1506, 2207, 2400), 33, alphabet="abc", stripes=(1024, 16384), facade=2)


def test_group_engine_runs_through_several_stripes_per_group():
    """On a corpus large enough the group kernel gives a group SEVERAL consecutive stripes of the index (span > 1: the stripe is
    chosen for lane-per-stripe engines): 256 MiB of 500-900-byte a/b lines at 1 KiB stripes - (a|b)*a(a|b){600} on 8 lanes x 3 words
    (span 2) and (a|b)*a(a|b){5000} on 32 lanes x 5 words (span 8) - bit for bit the wave-resident engine's result, the oracle's on
    the first lines."""
    import synth
    data = synth.corpus("ablong", 29, 256 << 20)
    dev = torch.from_numpy(data).cuda()
    corpus = rr.Corpus(dev, stripe=1024)
    head = data[:192 << 10]
    cut = int(np.flatnonzero(head == 10)[-1]) + 1
    for pattern in ("(a|b)*a(a|b){600}", "(a|b)*a(a|b){5000}"):
        r = rr.RRegex(pattern)
        assert r.engine_name == "nfa-group-cooperative"
        got = r.match_corpus_bits(corpus).clone()
        ref = rr.RRegex(pattern, rr.ENGINE_NFA_BLOCK).match_corpus_bits(corpus)
        assert torch.equal(got, ref), pattern
        want = OracleRegex(pattern).match_lines(head[:cut])
        assert (r.match_corpus(corpus).cpu().numpy()[:len(want)] == want).all(), pattern


def test_block_engine_512_lanes():
    """match_stripes_block_kernel at T = 512 (16385-32768 positions: 8 waves, the [2][T] / [2][2T] LDS exchange at twice
    the size) - never launched before round 3."""
    p = "(a|b)*a.{17000}"
    assert rr.RRegex(p).words_per_set == 532
    _coop_engine_case(p, rr.ENGINE_AUTO, "nfa-wave-resident", (0, 17001, 17002, 17003, 21000), 34,
                      stripes=(1024, 16384), per_length=1, facade=3, boundary_lines=3, boundary_deltas=(0,))


def test_sparse_live_set_engine():
    """SparseNfa (kernels_wave.hip): the state set kept as the list of its non-empty blocks of 2048 positions (the
    reference's sets are sparse: README.md:18-21, NFA.cc:77-85).  Rows 1 ... 16: a 604-position automaton (one block), 5003
    positions (blocks entered by the carry out of the block below), exception edges that reach across blocks, 17003 positions;
    texts on which one block is live and texts that fill every block."""
    _coop_engine_case("(a|b)*a(a|b){600}", rr.ENGINE_NFA_SPARSE, "nfa-wave-sparse", (0, 1, 600, 601, 602, 603, 700, 1300), 41)
    _coop_engine_case("(a|b)*a(a|b){5000}", rr.ENGINE_NFA_SPARSE, "nfa-wave-sparse", (0, 7, 90, 2047, 2048, 2049, 5001, 5002, 5003, 6100, 9000), 42,
                      stripes=(1024, 16384), boundary_lines=6)
    _coop_engine_case("((a|b)*a(a|b){2100}c|(b|c)*b(b|c){2500}a)*", rr.ENGINE_NFA_SPARSE, "nfa-wave-sparse",
                      (0, 3, 2102, 2103, 2502, 2503, 4605, 4606, 5207, 6400), 43, alphabet="abc", stripes=(1024, 16384), facade=2, boundary_lines=6)
    _coop_engine_case("(a|b)*a.{17000}", rr.ENGINE_NFA_SPARSE, "nfa-wave-sparse", (0, 17001, 17002, 17003, 21000), 34,
                      stripes=(16384,), per_length=1, facade=2, boundary_lines=2, boundary_deltas=(0,))


def test_small_automata_forced_onto_the_cooperative_engines():
    """a{1,300} (BASELINE configs[3]: 301 positions, a pure chain) forced onto the group and block engines over the synthetic
    `arepeat` corpus - batch entry and one-shot entry - and U2 / the email pattern likewise (exception edges, byte
    classes): bit-identical to the oracle."""
    import synth
    for kind, pattern, nbytes in (("arepeat", "a{1,300}", 192 << 10), ("url", U2, 96 << 10), ("email", EMAIL, 64 << 10)):
        data = synth.corpus(kind, 17, nbytes)
        want = OracleRegex(pattern).match_lines(data)
        dev = torch.from_numpy(data.copy()).cuda()
        for e in (rr.ENGINE_NFA_WAVE, rr.ENGINE_NFA_BLOCK, rr.ENGINE_NFA_SPARSE):
            r = rr.RRegex(pattern, e)
            for stripe in (0, 1024, 16384):
                got = r.match_corpus(rr.Corpus(dev, stripe=stripe)).cpu().numpy()
                bad = np.nonzero(got != want)[0]
                assert bad.size == 0, (kind, r.engine_name, stripe, "first bad line", int(bad[0]))
            bits, nlines = r.match_device_bits(dev)
            assert nlines == len(want) and (_bits_to_bytes(bits, nlines) == want).all(), (kind, r.engine_name, "one-shot")


def test_nfa_lane_engine_sixteen_words():
    """LineNfaEngine<16, ...> (385-512 positions in one lane's registers), all four builds: bare chain, + self loops and
    add-carry groups, + exception rows - never launched before round 3.  Batch, one-shot, extents, facade vs the oracle."""
    cases = [("a{1,450}", (0, 1, 2, 449, 450, 451, 452, 700), "a"),
             ("(a|b)*a(a|b){400}", (0, 1, 400, 401, 402, 403, 500, 900, 1700), "ab"),
             ("(ab|ba){1,120}", (0, 2, 3, 100, 238, 240, 242, 480), "ab"),
             ("b*a{1,440}", (0, 1, 5, 440, 441, 445, 600), "ab")]
    for pattern, lengths, alphabet in cases:
        r = rr.RRegex(pattern, rr.ENGINE_NFA)
        assert 13 <= r.words_per_set <= 16, (pattern, r.words_per_set)
        rng = random.Random(len(pattern))
        o = OracleRegex(pattern)
        lines = []
        for n in lengths:
            lines += ["".join(rng.choice(alphabet) for _ in range(n)) for _ in range(3)]
            lines += [alphabet[0] * n, (alphabet[-1] + alphabet[0] * n)[:max(n, 1)], ("ab" * n)[:n], ("ba" * n)[:n]]
        lines += ["".join(rng.choice(alphabet) for _ in range(rng.choice([3, 17, 60, 130]))) for _ in range(3000)]
        want = np.array([int(o.accepts(t)) for t in lines], dtype=np.uint8)
        assert 0 < want.sum() < len(want), pattern
        for tail in ("\n", ""):
            data = np.frombuffer(("\n".join(lines) + tail).encode(), dtype=np.uint8)
            dev = torch.from_numpy(data.copy()).cuda()
            for stripe in STRIPES:
                got = r.match_corpus(rr.Corpus(dev, stripe=stripe)).cpu().numpy()
                bad = np.nonzero(got != want)[0]
                assert bad.size == 0, (pattern, "stripe", stripe, "first bad line", int(bad[0]), len(lines[int(bad[0])]))
            bits, nlines = r.match_device_bits(dev)
            assert nlines == len(want) and (_bits_to_bytes(bits, nlines) == want).all(), (pattern, "one-shot")
        off = np.cumsum([0] + [len(t) for t in lines]).astype(np.int64)
        blob = torch.from_numpy(np.frombuffer("".join(lines).encode(), dtype=np.uint8).copy()).cuda()
        got = r.match_extents(blob, torch.from_numpy(off).cuda()).cpu().numpy()
        assert (got == want).all(), (pattern, "extents")
        for t in lines[:8]:
            assert (r.get_acceptance_iter(t).advance().value() is not None) == o.accepts(t)


def test_stripe_wise_items_against_the_oracle():
    """match_items_stripes_kernel (item ends from a bitmap, the plain table with an END OF ITEM column) checked directly
    against the ORACLE, item by item: an indexed batch (rrx_items) always takes the stripe-wise kernel, whatever its
    size.  trim 0 and 1, separators of any value, items longer than a stripe, 0x00 and bytes >= 0x80, '\n' inside items."""
    rng = np.random.default_rng(41)
    for pattern, alphabet in (("(a|b)*abb", b"ab"), ("(a|b|\n)*ab(a|\n)*", b"ab\n"), (EMAIL, b"ab1.@_Z")):
        o = OracleRegex(pattern.replace("\n", "x")) if "\n" in pattern else OracleRegex(pattern)
        r = rr.RRegex(pattern)
        assert r.engine == rr.ENGINE_DFA
        for trim in (0, 1):
            lens = rng.choice([1, 2, 3, 7, 19, 64, 300, 5000, 40_000], size=6000, p=[0.15, 0.15, 0.15, 0.2, 0.2, 0.1, 0.04, 0.009, 0.001])
            tot = int(lens.sum()) + trim * len(lens)
            text = np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), size=tot)].copy()
            dirty = rng.integers(0, tot, size=40)
            text[dirty] = rng.choice(np.array([0x00, 0x80, 0xff], dtype=np.uint8), size=len(dirty))
            ends = np.cumsum(lens + trim)
            if trim:
                text[ends - 1] = np.frombuffer(b"ab\n;\x00\xff", dtype=np.uint8)[rng.integers(0, 6, size=len(lens))]
            offs = np.concatenate([[0], ends]).astype(np.int64)
            items = rr.Items(torch.from_numpy(text).cuda(), torch.from_numpy(offs).cuda(), trim=trim)
            assert items.stripe_wise
            got = r.match_items(items).cpu().numpy()
            raw = text.tobytes()
            want = np.array([int(o.accepts(raw[offs[i]:offs[i + 1] - trim].replace(b"\n", b"x") if "\n" in pattern else raw[offs[i]:offs[i + 1] - trim]))
                             for i in range(len(lens))], dtype=np.uint8)
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, (pattern[:20], "trim", trim, "first bad item", int(bad[0]), int(lens[bad[0]]))
            assert 0 < want.sum() < len(want)


def _bits_to_bytes(words, n):
    w = words.cpu().numpy().view(np.uint32)
    return ((w[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.uint8).reshape(-1)[:n]


def test_one_shot_entry_reads_the_text_once_and_agrees_with_the_indexed_path():
    """rrx_match_device: no rrx_corpus, the newline index is a by-product of the match (stride-2 table engine) - against
    the oracle and against the indexed path: ragged lines, empty lines, lines longer than a stripe, bytes >= 0x80 and 0x00
    (the one-pass kernel cannot rely on an index pass to route such corpora elsewhere), with and without a final '\n',
    sizes around stripe and workgroup multiples; an engine without a stride-2 table takes the two-pass fallback."""
    rng = np.random.default_rng(51)
    p = "(a|b)*abb(a|b)*"
    o = OracleRegex(p)
    alphabet = np.frombuffer(b"ab\n", dtype=np.uint8)
    sizes = [1, 2, 15, 16, 17, 127, 128, 129, 2047, 2048, 2049, 4096 + 5, 1024 * 2048, 1024 * 2048 + 2048 + 77, 5_000_003]
    for n in sizes:
        for variant in ("plain", "dirty", "long"):
            probs = [0.47, 0.47, 0.06] if variant != "long" else [0.4999, 0.4999, 0.0002]
            data = alphabet[rng.choice(3, size=n, p=probs)].copy()
            if variant == "dirty" and n > 8:
                idx = rng.integers(0, n, size=max(1, n // 500))
                data[idx] = rng.choice(np.array([0x00, 0x80, 0xff, 0xc3], dtype=np.uint8), size=len(idx))
            for last in (10, 97):
                data[-1] = last
                want = o.match_lines(data)
                dev = torch.from_numpy(data.copy()).cuda()
                for e in (rr.ENGINE_AUTO, rr.ENGINE_NFA):
                    r = rr.RRegex(p, e)
                    bits, nlines = r.match_device_bits(dev)
                    assert nlines == len(want), (n, variant, last, r.engine_name)
                    got = _bits_to_bytes(bits, nlines)
                    bad = np.nonzero(got != want)[0]
                    assert bad.size == 0, (n, variant, last, r.engine_name, "first bad line", int(bad[0]))
                if n > 100000:
                    break
    # extremes: nothing but newlines (one string per byte: every lane's stream fills its slab), no newline at all (one
    # string that every lane but the first merely follows), one-byte lines
    for blob in (b"\n" * ((1 << 20) + 3), b"ab" * 300_000, b"a\n" * 400_001, b"abb\n" + b"b" * 70_000 + b"\nabb"):
        data = np.frombuffer(blob, dtype=np.uint8).copy()
        want = o.match_lines(data)
        for e in (rr.ENGINE_AUTO, rr.ENGINE_DFA, rr.ENGINE_NFA):
            bits, nlines = rr.RRegex(p, e).match_device_bits(torch.from_numpy(data.copy()).cuda())
            assert nlines == len(want) and (_bits_to_bytes(bits, nlines) == want).all(), (len(blob), e)
    # a bitmap that is too small is reported, not overrun
    data = np.frombuffer(b"a\n" * 5000, dtype=np.uint8).copy()
    with pytest.raises(rr.RRegexError, match="too small"):
        rr.RRegex("a").match_device_bits(torch.from_numpy(data).cuda(), cap_lines=1024)
    # BASELINE corpora, one-shot == indexed path
    import synth
    for kind, pattern, nbytes in (("url", U2, 64 << 20), ("kwlines", K1000, 16 << 20), ("arepeat", "a{1,300}", 16 << 20)):
        host = synth.corpus(kind, 13, nbytes)
        dev = torch.from_numpy(host).cuda()
        r = rr.RRegex(pattern)
        assert r.engine_name == "dfa-stride2-table"
        bits, nlines = r.match_device_bits(dev)
        corpus = rr.Corpus(dev)
        assert nlines == corpus.num_lines
        assert torch.equal(bits, r.match_corpus_bits(corpus)), kind


def test_block_cooperative_engine_beyond_4096_positions():
    """An automaton with more than 8192 positions (9003: no table form exists, beyond the group engine) compiles to the
    wave-resident engine and matches the oracle: batch kernel, extents kernel, iterator facade."""
    rng = random.Random(78)
    p = "(a|b)*a(a|b){9000}"
    r = rr.RRegex(p)
    assert r.engine_name == "nfa-wave-resident"
    o = OracleRegex(p)
    lines = ["", "a", "a" + "b" * 9000, "b" + "b" * 9000, "ab" * 100 + "a" + "a" * 9000, "a" * 8999, "a" * 9002]
    for n in (9001, 9002, 9600, 14000):
        lines.append("".join(rng.choice("ab") for _ in range(n)))
    data = ("\n".join(lines)).encode()
    want = o.match_lines(np.frombuffer(data, dtype=np.uint8))
    assert 0 < want.sum() < len(want)
    dev = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    for stripe in (1024, 16384):
        got = r.match_corpus(rr.Corpus(dev, stripe=stripe)).cpu().numpy()
        assert (got == want).all(), stripe
    off = np.cumsum([0] + [len(t) for t in lines]).astype(np.int64)
    blob = torch.from_numpy(np.frombuffer("".join(lines).encode(), dtype=np.uint8).copy()).cuda()
    got = r.match_extents(blob, torch.from_numpy(off).cuda()).cpu().numpy()
    assert list(got) == [int(x) for x in want]
    for t in lines[:5]:
        assert (r.get_acceptance_iter(t).advance().value() is not None) == o.accepts(t)
    # (lines of b only keep a single position alive: every 4096-position block but the first is empty and skips its step;
    # lines of a only fill the set: both sides of the per-wave skip are in `lines`)


def test_bounded_repeat_of_twenty_thousand_copies():
    """a{1,20000}: 59,999 reference states (under the front end's 65,536), the family the round-3 pipeline did not finish
    compiling (Parser.cpp:123-141 makes n^2/2 edges by plain copies).  Closed-form answer - a line is accepted
    iff it is 1..20000 bytes of `a` - through the batch kernel, the one-shot entry, explicit items and the iterator facade."""
    import time
    rng = random.Random(91)
    t0 = time.time()
    r = rr.RRegex("a{1,20000}")
    assert time.time() - t0 < 5.0
    assert r.states_n == 59999 and r.useful_states == 20001 and r.engine_name == "nfa-wave-resident"
    lines = [b"", b"a", b"aa", b"b", b"a" * 19999, b"a" * 20000, b"a" * 20001, b"a" * 26000, b"a" * 9000 + b"b" + b"a" * 9000,
             b"a" * 19999 + b"b", b"b" + b"a" * 100, b"a" * 12345 + b"\x80", b"a" * 40000]
    for _ in range(40):
        n = rng.choice([rng.randint(1, 300), rng.randint(15000, 25000), rng.randint(19990, 20010)])
        ln = bytearray(b"a" * n)
        if rng.random() < 0.3:
            ln[rng.randrange(n)] = rng.choice(b"bA \x00")
        lines.append(bytes(ln))
    want = np.array([1 if 1 <= len(ln) <= 20000 and ln == b"a" * len(ln) else 0 for ln in lines], dtype=np.uint8)
    assert 10 < want.sum() < len(want) - 10
    for tail in (b"\n", b""):
        data = np.frombuffer(b"\n".join(lines) + tail, dtype=np.uint8)
        dev = torch.from_numpy(data.copy()).cuda()
        for stripe in (1024, 16384, 0):
            got = r.match_corpus(rr.Corpus(dev, stripe=stripe)).cpu().numpy()
            assert got.shape == want.shape and (got == want).all(), (stripe, np.nonzero(got != want)[0][:5])
        bits, n = r.match_device_bits(dev)
        assert n == len(lines)
        got = ((bits.cpu().numpy().view(np.uint32)[np.arange(n) >> 5] >> (np.arange(n) & 31)) & 1).astype(np.uint8)
        assert (got == want).all(), np.nonzero(got != want)[0][:5]
    off = np.cumsum([0] + [len(t) for t in lines]).astype(np.int64)
    blob = torch.from_numpy(np.frombuffer(b"".join(lines), dtype=np.uint8).copy()).cuda()
    got = r.match_extents(blob, torch.from_numpy(off).cuda()).cpu().numpy()
    assert (got == want).all(), np.nonzero(got != want)[0][:5]
    for ln, w in list(zip(lines, want))[:14]:
        m = r.get_acceptance_iter(ln).advance().value()
        assert (m is not None) == bool(w), len(ln)
        if m is not None:
            assert m.start == 0 and m.end == len(ln)
    # (ab){1,5000} and (abc|de){1,3000} determinise (10002 / 12002 table states: the table in HBM/L2): same closed forms
    for p, unit in (("(ab){1,5000}", [b"ab"]), ("(abc|de){1,3000}", [b"abc", b"de"])):
        r = rr.RRegex(p)
        top = 5000 if unit == [b"ab"] else 3000
        ls, ws = [], []
        for k in [0, 1, 2, top - 1, top, top + 1, top + 50] + [rng.randint(1, top + 200) for _ in range(12)]:
            ln = b"".join(rng.choice(unit) for _ in range(k))
            ls.append(ln); ws.append(1 if 1 <= k <= top else 0)
            if k:
                ls.append(ln[:-1]); ws.append(int(_units(ln[:-1], unit, top)))
        data = np.frombuffer(b"\n".join(ls) + b"\n", dtype=np.uint8)
        got = r.match_corpus(rr.Corpus(torch.from_numpy(data.copy()).cuda())).cpu().numpy()
        assert list(got) == ws, (p, r.engine_name)


def _units(ln, unit, top):
    """ln is a concatenation of 1..top words of `unit` (words are prefix-free here)."""
    k = 0
    while ln:
        for u in unit:
            if ln.startswith(u):
                ln = ln[len(u):]
                k += 1
                break
        else:
            return False
    return 1 <= k <= top


def _sharded_rank(rank, world, port, q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tools"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import numpy as np
    import torch
    import torch.distributed as dist
    import roaringregex_amd as rr
    import synth
    from patterns import U2
    from roaringregex_amd.shard import match_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)                                  # a one-GPU box: both ranks share the card, gloo carries the results
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = synth.corpus("url", 9, 24 << 20, threads=2)[:-11]       # an unterminated last line on purpose
        r = rr.RRegex(U2)
        whole = match_sharded(lambda shard: r.match_corpus(rr.Corpus(torch.from_numpy(np.ascontiguousarray(shard)).cuda())), data, rank, world)
        q.put((rank, synth.fnv1a(whole), int(whole.size), int(whole.sum())))
    finally:
        dist.destroy_process_group()


def test_one_corpus_sharded_with_the_hip_matcher():
    """SURVEY 8(e) to the letter: ONE corpus cut into line-aligned byte ranges (roaringregex_amd.shard), every rank scans its range
    with RRegex.match_corpus, the accept vectors concatenated in rank order - at world 1 in this process, at world 2 in two
    processes over gloo (both on GPU 0): equal to the unsharded HIP result and, on a sample, to the oracle."""
    import socket
    import synth
    import torch.multiprocessing as mp
    from roaringregex_amd.shard import match_sharded
    data = synth.corpus("url", 9, 24 << 20, threads=2)[:-11]
    r = rr.RRegex(U2)
    unsharded = r.match_corpus(rr.Corpus(torch.from_numpy(data.copy()).cuda())).cpu().numpy()
    one = match_sharded(lambda shard: r.match_corpus(rr.Corpus(torch.from_numpy(np.ascontiguousarray(shard)).cuda())), data, 0, 1)
    assert one.shape == unsharded.shape and (one == unsharded).all()
    head = data[:1 << 20]
    cut = int(np.flatnonzero(head == 10)[-1]) + 1
    want = OracleRegex(U2).match_lines(head[:cut])
    assert (unsharded[:len(want)] == want).all() and 0 < want.sum() < len(want)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_rank, args=(k, 2, port, q)) for k in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, h, n, acc in res:
        assert (h, n, acc) == (synth.fnv1a(unsharded), unsharded.size, int(unsharded.sum())), rank


def test_match_host_pipeline_equals_resident_path():
    """rrx_match_host on an input larger than its 256 MiB chunk: the chunked, double-buffered upload must give
    exactly the accept vector of the device-resident path (and of the oracle on a sampled chunk)."""
    import synth
    n = (600 << 20) + 12345
    host = synth.corpus("url", 9, n)
    r = rr.RRegex(U2)
    got = r.match_host(host)
    want = r.match_corpus(rr.Corpus(torch.from_numpy(host).cuda())).cpu().numpy()
    assert got.shape == want.shape and (got == want).all()
    o = OracleRegex(U2)
    piece = host[:1 << 20]
    assert (got[:int((piece == 10).sum())] == o.match_lines(piece)).all()


def test_extents_with_unaligned_items_of_many_lengths():
    """The extents kernel walks items with 16-byte loads between an unaligned head and tail: items of every length
    0..80 at every alignment, plus a few long ones, with dead bytes sprinkled in."""
    rng = np.random.default_rng(12)
    items = []
    for n in list(range(0, 81)) + [255, 256, 257, 1000, 4097]:
        body = np.frombuffer(b"ab\n", dtype=np.uint8)[rng.integers(0, 3, size=n)].copy()
        if n and rng.random() < 0.15:
            body[rng.integers(0, n)] = rng.choice([0, 0x80, 0xff])
        items.append(body.tobytes())
    blob = b"".join(items)
    off = np.cumsum([0] + [len(i) for i in items]).astype(np.int64)
    data = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).cuda()
    offs = torch.from_numpy(off).cuda()
    p = "(a|b|\n)*ab(a|\n)*"
    o = OracleRegex("(a|b|x)*ab(a|x)*")        # the oracle API is NUL/newline-agnostic: model '\n' as 'x'
    want = []
    for it in items:
        t = it.replace(b"\n", b"x")
        want.append(int(o.accepts(t)))
    for r in engines_for(p, small=True):
        got = r.match_extents(data, offs).cpu().numpy()
        assert list(got) == want, r.engine_name


def test_explicit_items_stripe_wise():
    """rrx_match_extents on large batches (an offsets array over one buffer) takes the stripe-wise kernel: item ends from a
    bitmap built from the offsets, the plain table with an end-of-item column.  Checked against the lane-per-item kernel
    (the same call on pieces of 50000 items, below the threshold) and, where the items are the lines of a corpus, against
    the batch kernel: trim 1 with '\n' and with ';' as the separator, trim 0 (no separator at all), items that hold a real
    '\n', 0x00 and bytes >= 0x80, empty items (trim 0: the batch falls back), a pattern that survives a '\n'."""
    import synth
    n = 24 << 20

    def pieces(r, dev, off, trim):
        out = []
        for lo in range(0, off.numel() - 1, 50000):
            out.append(r.match_extents(dev, off[lo:lo + 50001].contiguous(), trim=trim))
        return torch.cat(out)

    for kind, pattern in (("url", U2), ("email", EMAIL)):
        host = synth.corpus(kind, 9, n).copy()
        rng = np.random.default_rng(5)
        spots = rng.integers(0, n, 200)
        host[spots[:100]] = 0
        host[spots[100:]] = 0xC3
        host[host == 10][:0]                                            # (separators stay where they are)
        dev = torch.from_numpy(host).cuda()
        nl = torch.nonzero(dev == 10).flatten()
        off = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), nl + 1]).contiguous()
        for engine in (rr.ENGINE_AUTO, rr.ENGINE_DFA):
            r = rr.RRegex(pattern, engine)
            want = r.match_corpus(rr.Corpus(dev))[:off.numel() - 1].bool()
            # trim 1, separator '\n'
            got = r.match_extents(dev, off, trim=1).bool()
            assert torch.equal(got, want), (kind, r.engine_name, "trim 1")
            # a batch that starts in the middle of the buffer: at a 16-byte boundary (stripe-wise) and not (lane per item)
            o = off.cpu().numpy()
            k_al = int(np.nonzero((o[1000:2000] % 16) == 0)[0][0]) + 1000
            k_un = int(np.nonzero((o[1000:2000] % 16) != 0)[0][0]) + 1000
            for k in (k_al, k_un):
                got = r.match_extents(dev, off[k:].contiguous(), trim=1).bool()
                assert torch.equal(got, want[k:]), (kind, r.engine_name, "from item", k)
            # trim 1, separator ';' and real '\n' bytes inside some items (they reject their items)
            h2 = host.copy(); h2[h2 == 10] = ord(";")
            inside = np.nonzero(h2 != ord(";"))[0][::100003]
            h2[inside] = 10
            d2 = torch.from_numpy(h2).cuda()
            got = r.match_extents(d2, off, trim=1).bool()
            ref = pieces(r, d2, off, 1).bool()
            assert torch.equal(got, ref), (kind, r.engine_name, "trim 1, ';'")
            assert int((got != want).sum()) <= len(inside)
            # trim 0: the same items without any separator
            keep = host != 10
            h0 = host[keep]
            lens = np.diff(np.concatenate([[0], np.nonzero(host == 10)[0] + 1])) - 1
            off0 = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).cuda()
            d0 = torch.from_numpy(h0.copy()).cuda()
            got = r.match_extents(d0, off0, trim=0).bool()
            if int((torch.from_numpy(lens) == 0).sum()):                 # empty items: the batch went the other way; same answer
                pass
            assert torch.equal(got, want), (kind, r.engine_name, "trim 0")
            # trim 0 without empty items (so that the stripe-wise kernel is the one that runs)
            nz = np.nonzero(lens > 0)[0]
            if len(nz) != len(lens):
                lens1 = lens[nz]
                off1 = torch.from_numpy(np.concatenate([[0], np.cumsum(lens1)]).astype(np.int64)).cuda()
                got = r.match_extents(d0, off1, trim=0).bool()
                assert torch.equal(got, want[torch.from_numpy(nz).cuda()]), (kind, r.engine_name, "trim 0, no empty items")
    # items from 1 byte to 200 KB (longer than many stripes: lanes whose stripe holds no item end at all), separators of any
    # value - also bytes the pattern matches -, a buffer whose last stripe is partial
    rng = np.random.default_rng(23)
    lens = rng.choice([1, 2, 5, 30, 300, 5000, 200_000], size=70_000, p=[0.2, 0.2, 0.25, 0.25, 0.09, 0.009, 0.001])
    r = rr.RRegex("(a|b)*abb")
    for trim in (0, 1):
        tot = int(lens.sum()) + trim * len(lens)
        text = np.frombuffer(b"ab", dtype=np.uint8)[rng.integers(0, 2, size=tot)].copy()
        ends = np.cumsum(lens + trim)
        if trim:
            text[ends - 1] = np.frombuffer(b"ab\nx", dtype=np.uint8)[rng.integers(0, 4, size=len(lens))]
        offs = torch.from_numpy(np.concatenate([[0], ends]).astype(np.int64)).cuda()
        dev = torch.from_numpy(text).cuda()
        assert torch.equal(r.match_extents(dev, offs, trim=trim), pieces(r, dev, offs, trim)), ("long items", trim)
        # the same batch indexed once (rrx_items) and matched by several patterns
        items = rr.Items(dev, offs, trim=trim)
        assert items.num_items == len(lens) and items.stripe_wise
        for pat in ("(a|b)*abb", "a*b*", "(ab|ba)*a?", "[^x]*"):
            rp = rr.RRegex(pat)
            assert torch.equal(rp.match_items(items), pieces(rp, dev, offs, trim)), ("items handle", pat, trim)
        nfa = rr.RRegex("(a|b)*abb", rr.ENGINE_NFA)                     # an engine without the stripe-wise form: lane per item
        assert torch.equal(nfa.match_items(items), pieces(r, dev, offs, trim))
    # a batch with an empty item at trim 0 is indexed, found degenerate, and matched lane per item
    o2 = torch.tensor([0, 3, 3, 8], dtype=torch.int64, device="cuda")
    d2 = torch.from_numpy(np.frombuffer(b"abbaaabb", dtype=np.uint8).copy()).cuda()
    it2 = rr.Items(d2, o2, trim=0)
    assert not it2.stripe_wise and r.match_items(it2).tolist() == [1, 0, 1]
    # a pattern for which '\n' is an ordinary, matchable byte: same answers as on pieces
    host = synth.corpus("email", 4, n)
    dev = torch.from_numpy(host).cuda()
    nl = torch.nonzero(dev == 10).flatten()
    off = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), nl + 1]).contiguous()
    r = rr.RRegex("[^@]*")
    assert torch.equal(r.match_extents(dev, off, trim=0), pieces(r, dev, off, 0))
    # ... and with a separator behind every item (trim 1: the stride-2 items table, in which '\n' is a byte like any other and only
    # the marked byte ends an item): ';' as the separator, real '\n' bytes inside some items - accepted by this pattern
    h2 = host.copy(); h2[h2 == 10] = ord(";")
    inside = np.nonzero(h2 != ord(";"))[0][::50021]
    h2[inside] = 10
    d2 = torch.from_numpy(h2).cuda()
    for pat in ("[^@]*", ".*", "[^;]*"):
        rp = rr.RRegex(pat)
        got = rp.match_extents(d2, off, trim=1)
        assert torch.equal(got, pieces(rp, d2, off, 1)), pat
        assert torch.equal(rp.match_items(rr.Items(d2, off, trim=1)), got), pat
        rp.set_items_stride2(False)                                     # the byte-stride items kernel on the same batch
        assert torch.equal(rp.match_extents(d2, off, trim=1), got), pat
    o = OracleRegex("[^@]*")
    got = rr.RRegex("[^@]*").match_extents(d2, off, trim=1).cpu().numpy()
    offs = off.cpu().numpy()
    for i in list(range(0, 2000)) + [int(np.searchsorted(offs, p, side="right")) - 1 for p in inside[:50]]:
        item = h2[offs[i]:offs[i + 1] - 1].tobytes()
        assert got[i] == (1 if o.accepts(item) else 0), (i, item)


def test_explicit_items_inside_a_far_larger_allocation():
    """rrx_match_extents takes what is left of the allocation behind d_bytes as the bound of its index only while that is plausible for
    the batch (128 bytes per item); a batch inside a memory pool has its real extent read back instead (one synchronisation).  Same answers as on a buffer
    of the batch's own size, from the front and from the middle of a 10 GiB allocation."""
    import synth
    n = 24 << 20
    host = synth.corpus("url", 12, n)
    last = int(np.nonzero(host == 10)[0][-1]) + 1
    host = host[:last]
    off = torch.from_numpy(np.concatenate([[0], np.nonzero(host == 10)[0] + 1]).astype(np.int64)).cuda()
    small = torch.from_numpy(host).cuda()
    r = rr.RRegex(U2)
    want = r.match_extents(small, off, trim=1)
    assert int(want.sum()) > 1000
    def rate(data):
        out = torch.empty(off.numel() - 1, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            r.match_extents(data, off, trim=1, out=out)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            r.match_extents(data, off, trim=1, out=out)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / 5

    alone_ms = rate(small)
    pool = torch.empty(10 << 30, dtype=torch.uint8, device="cuda")
    for at in (0, 6 << 30):                                                  # (6 GiB in: 4 GiB left behind the batch)
        pool[at:at + last] = small
        got = r.match_extents(pool[at:at + last], off, trim=1)
        assert torch.equal(got, want), at
        # sized from the pool's tail the batch ran on two workgroups' worth of stripes and took 512 MiB of scratch;
        # sized from its own extent it runs as it does alone (one synchronisation more)
        assert rate(pool[at:at + last]) < 2.0 * alone_ms + 0.1, (at, alone_ms)
    del pool
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------ one long string
def test_one_long_string_is_stepped_in_parallel_chunks():
    """regex.h:156-159 consumes ONE string; rrx_match_string cuts a long one into chunks, steps every chunk from every
    table state and composes the chunk maps.  Lengths straddle the 32 KiB threshold, the chunk size and one, two and
    three composition levels; '\n' is an ordinary byte; a byte >= 0x80 rejects."""
    rng = np.random.default_rng(21)
    ab = np.frombuffer(b"ab", dtype=np.uint8)
    cases = []
    for n in (1023, 1024, 1025, 2047, 2048, 2049, 3000, 8191, 32 * 1024 - 1, 32 * 1024, 32 * 1024 + 1, 100_003, 262143, 262145, (1 << 20) + 17, 5 * (1 << 20) + 1023, 20 * (1 << 20) + 5):
        body = ab[rng.integers(0, 2, size=n)]
        for tail in (b"abb", b"aba"):
            t = body.copy()
            t[-3:] = np.frombuffer(tail, dtype=np.uint8)
            cases.append(("(a|b)*abb", t))
    text = np.frombuffer(b"the quick brown fox\n", dtype=np.uint8)
    big = np.tile(text, 60000)                                  # 1.2 MB with '\n' inside
    hit = big.copy(); hit[777_777:777_780] = np.frombuffer(b"abc", dtype=np.uint8)
    cases += [(".*abc.*", big), (".*abc.*", hit), ("abc", hit), ("(the quick brown fox.)*", big), ("(the quick brown fox.)*", big[:-1])]
    high = hit.copy(); high[999_999] = 0xC3
    cases.append((".*abc.*", high))
    url = np.frombuffer(b"https://www.example.com/" + b"a/b-c_d.e" * 30000 + b"?q=1", dtype=np.uint8)
    cases += [(U2, url), (U2, url[:-4]), (U2, np.concatenate([url, np.frombuffer(b" x", dtype=np.uint8)]))]
    # an automaton that counts does not forget where it started: the chunks keep more distinct states than the
    # convergence path has slots and fall back to stepping every state through the whole chunk
    runs = np.frombuffer((b"a" * 37 + b"b") * 2500, dtype=np.uint8)
    broken = runs.copy(); broken[50_000:50_120] = ord("a")
    cases += [("(a{1,100}b)*", runs), ("(a{1,100}b)*", broken), ("(a{1,100}b)*", runs[:-1])]
    oracles = {}
    for pattern, t in cases:
        o = oracles.setdefault(pattern, OracleRegex(pattern))
        want = o.accepts(t.tobytes())
        dev = torch.from_numpy(np.array(t, copy=True)).cuda()
        for e in (rr.ENGINE_AUTO, rr.ENGINE_DFA, rr.ENGINE_NFA):       # (NFA engines: chunk relations, lane = (chunk, start position))
            r = rr.RRegex(pattern, e)
            assert r.match_string(dev) == want, (pattern[:30], len(t), r.engine_name)
    # the iterator facade takes the same path for a long host string
    r = rr.RRegex("(a|b)*abb")
    s = (b"ab" * 100000) + b"abb"
    it = r.get_acceptance_iter(s).advance()
    m = it.value()
    assert m is not None and (m.start, m.end) == (0, len(s))
    assert r.get_acceptance_iter(s + b"a").advance().value() is None


# ------------------------------------------------------------------------------------------ search
def _check_search(pattern, data, want_start, want_end, stripes=(0,)):
    dev = torch.from_numpy(np.array(np.frombuffer(data, dtype=np.uint8), copy=True)).cuda()
    r = rr.RRegex(pattern)
    for stripe in stripes:
        corpus = rr.Corpus(dev, stripe=stripe)
        assert corpus.num_lines == len(want_start)
        s, e = r.search_corpus(corpus)
        s, e = s.cpu().numpy(), e.cpu().numpy()
        bad = np.nonzero((s != want_start) | (e != want_end))[0]
        assert bad.size == 0, (pattern[:40], "stripe", corpus.stripe, "line", int(bad[0]), (int(s[bad[0]]), int(e[bad[0]])),
                               "want", (int(want_start[bad[0]]), int(want_end[bad[0]])))


def _oracle_confirms_search(pattern, lines, first=None, every=None, seed=1, starts_per_match=10):
    """Search results against the ORACLE's own compile of the pattern, not against a replay of the device's tables.  With
    `contains` = the oracle's automaton of `.*(p).*` (`.` covers all 128 codes, Parser.cpp:106-109) and `o` that of p:
      * a line reported without a match: `contains` rejects the whole line (no substring is accepted);
      * a match [s, e) searched from position b: o accepts line[s:e]; `contains` rejects line[b:e-1] (no match ends earlier - a
        complete check, one oracle run); no s' in a sample of [b, s) (always b, s-1, s-2) has line[s':e] accepted.
    `first`: one (s, e) per line; `every`: the list of all matches per line, each searched from the end of the one before."""
    rng = random.Random(seed)
    o, contains = OracleRegex(pattern), OracleRegex(".*(" + pattern + ").*")

    def confirm(ln, b, s, e):
        assert b <= s <= e <= len(ln), (pattern[:30], b, s, e)
        assert o.accepts(ln[s:e].decode("latin-1")), (pattern[:30], "not a match", s, e)
        if e - 1 >= b:
            assert not contains.accepts(ln[b:e - 1].decode("latin-1")), (pattern[:30], "a match ends before", e)
        for s2 in {b, s - 1, s - 2} | {rng.randrange(b, s) for _ in range(starts_per_match) if s > b}:
            if b <= s2 < s:
                assert not o.accepts(ln[s2:e].decode("latin-1")), (pattern[:30], "an earlier start", s2, "for", (s, e))

    for i, ln in enumerate(lines):
        if first is not None:
            s, e = first[i]
            if e < 0:
                assert not contains.accepts(ln.decode("latin-1")), (pattern[:30], "line", i, "has a match")
            else:
                confirm(ln, 0, s, e)
        if every is not None:
            b = 0
            for s, e in every[i]:
                confirm(ln, b, s, e)
                b = e if e > s else e + 1                       # (an empty match: the search moves on by one byte)
            if b <= len(ln):
                assert not contains.accepts(ln[b:].decode("latin-1")), (pattern[:30], "line", i, "has a further match after", b)


def test_search_corpus_short_lines_against_the_oracle():
    """rrx_search_corpus: per line the accepted substring with the smallest end, then the smallest start.  The oracle
    finds it by brute force with the reference's whole-string acceptance (short lines only)."""
    rng = random.Random(77)
    pats = ["ab+c", "a*", "(a|b)*abb", "[0-9]+\\.[0-9]+", "x?y?z?", "k(1|10|100)", "a{2,4}b", ".*c", "c.*", "[^a]b", EMAIL, U2]
    while len(pats) < 40:
        p = random_pattern(rng)
        try:
            if OracleRegex(p).states_n <= 120:
                pats.append(p)
        except OracleError:
            pass
    for p in pats:
        o = OracleRegex(p)
        alphabet = "abcxk01.d@yz" if p not in (EMAIL, U2) else "ab1.@:/hftps"
        lines = ["".join(rng.choice(alphabet) for _ in range(rng.choice([0, 0, 1, 2, 5, 9, 14, 22]))).encode() for _ in range(3000)]
        lines[5] = b"\x80ab" + lines[5]
        lines[9] = lines[9] + b"\xc3\xa9" + lines[10]
        for tail in (b"\n", b""):
            data = b"\n".join(lines) + tail
            ws, we = o.search_lines(data)
            _check_search(p, data, ws, we, stripes=(1024, 4096))
    # degenerate inputs
    for data in (b"\n", b"\n\n\n", b"a", b"abc\n"):
        ws, we = OracleRegex("ab?").search_lines(data)
        _check_search("ab?", data, ws, we)
    r = rr.RRegex("ab")
    s, e = r.search_corpus(rr.Corpus(torch.empty(0, dtype=torch.uint8, device="cuda")))
    assert s.numel() == 0 and e.numel() == 0


def test_search_first_match_where_a_chunk_overflows_the_staging_array():
    """First match per line where a 16-KiB chunk holds more lines than the kernel's staging array: (1) a corpus of short lines
    throughout - the launcher picks the build that takes the lines in windows; (2) a corpus of long lines with one dense stretch of
    3-byte lines - the plain build, whose overflowing chunks take the wave-wide event loop and write the lines beyond the array one
    by one.  Against the CPU replay of the two search tables (pinned to the oracle by tests/test_lowering.py), the short lines against
    the oracle's brute force as well."""
    from program_replay import SearchReplay
    rng = random.Random(97)
    for pattern in ("ab+c", "k(1|10|100)", EMAIL):
        r = rr.RRegex(pattern)
        rep = SearchReplay(r.program(rr.PROGRAM_SEARCH_FWD), r.program(rr.PROGRAM_SEARCH_REV))
        alphabet = "abck01.@ "
        short = [bytes(rng.choice(alphabet.encode()) for _ in range(rng.choice([0, 1, 2, 3, 3, 4, 6]))) for _ in range(40000)]
        long_ = [bytes(rng.choice(alphabet.encode()) for _ in range(rng.randint(150, 400))) for _ in range(3000)]
        for name, lines in (("short throughout", short), ("a dense stretch", long_[:1500] + short[:12000] + long_[1500:])):
            data = np.frombuffer(b"\n".join(lines) + b"\n", dtype=np.uint8)
            want = [rep.search(ln) for ln in lines]
            corpus = rr.Corpus(torch.from_numpy(data.copy()).cuda())
            assert corpus.num_lines == len(lines)
            s, e = r.search_corpus(corpus)
            got = list(zip(s.cpu().tolist(), e.cpu().tolist()))
            bad = [i for i, (g, w) in enumerate(zip(got, want)) if g != w]
            assert not bad, (pattern, name, bad[:3], got[bad[0]], want[bad[0]])
        ws, we = OracleRegex(pattern).search_lines(b"\n".join(short[:4000]) + b"\n")
        assert [(int(a), int(b)) for a, b in zip(ws, we)] == [rep.search(ln) for ln in short[:4000]], pattern


def test_search_corpus_long_lines_across_stripes():
    """Lines far longer than a stripe (matches found after, at and across stripe boundaries; the walk back to the match
    start crosses them too), checked with the CPU replay of the same two tables, which tests/test_lowering.py pins to
    the oracle."""
    from program_replay import SearchReplay
    rng = random.Random(5)
    for p in ("ab+c", "[0-9]+\\.[0-9]+", "b(a|c)*d", U2):
        r = rr.RRegex(p)
        rep = SearchReplay(r.program(rr.PROGRAM_SEARCH_FWD), r.program(rr.PROGRAM_SEARCH_REV))
        needle = {"ab+c": b"abbbbc", "[0-9]+\\.[0-9]+": b"12345.678", "b(a|c)*d": b"b" + b"ac" * 700 + b"d", U2: b"http://www.example.com/a/b"}[p]
        lines = []
        for n in (0, 1, 1000, 1017, 1024, 1030, 2048, 3000, 4090, 4096, 5000, 9000, 17000):
            filler = bytes(rng.choice(b"xyz ") for _ in range(n))
            lines += [filler + needle + b" tail", filler, needle + filler, filler[: n // 2] + needle[:3] + filler[n // 2:] + needle]
        data = b"\n".join(lines)
        want = [rep.search(ln) for ln in lines]
        _oracle_confirms_search(p, lines, first=want)            # the replay's answers are the oracle's
        ws = np.array([w[0] for w in want], dtype=np.int32)
        we = np.array([w[1] for w in want], dtype=np.int32)
        _check_search(p, data, ws, we, stripes=STRIPES)


def test_search_all_matches_per_line():
    """rrx_search_all_count / _fill: every lazy match of every line, left to right (after a match the search continues
    at its end, one byte further after an empty match).  Short lines against the oracle's brute force, long lines
    (matches on both sides of stripe boundaries) against the CPU replay of the two search tables."""
    from program_replay import SearchReplay
    rng = random.Random(13)
    pats = ["ab+c", "a*", "a?", "(a|b)*abb", "[0-9]+", "x?y?z?", "k(1|10|100)", "a{2,4}b", "[^a]b", EMAIL]
    while len(pats) < 26:
        p = random_pattern(rng)
        try:
            if OracleRegex(p).states_n <= 120:
                pats.append(p)
        except OracleError:
            pass

    def run(pattern, data):
        dev = torch.from_numpy(np.array(np.frombuffer(data, dtype=np.uint8), copy=True)).cuda()
        r = rr.RRegex(pattern)
        corpus = rr.Corpus(dev, stripe=1024)
        cnt, first, st, en = r.search_all(corpus)
        # the one-call entry (one pass, decoupled look-back) must give the same CSR arrays, also when the first call's
        # arrays were too small (cap = 1: counted but not written, then repeated with the exact size)
        for cap in (None, 1):
            f2, s2, e2 = r.search_all_fused(corpus, cap=cap)
            assert f2.numel() == cnt.numel() + 1 and int(f2[-1]) == st.numel(), (pattern, cap)
            assert torch.equal(f2[:-1], first) and torch.equal(s2, st) and torch.equal(e2, en), (pattern, cap)
        return cnt.cpu().numpy(), first.cpu().numpy(), st.cpu().numpy(), en.cpu().numpy()

    for p in pats:
        o = OracleRegex(p)
        alphabet = "abcxk01.d@yz" if p != EMAIL else "ab1.@"
        lines = ["".join(rng.choice(alphabet) for _ in range(rng.choice([0, 0, 1, 2, 5, 9, 14, 22]))).encode() for _ in range(2500)]
        for tail in (b"\n", b""):
            data = b"\n".join(lines) + tail
            wc, ws, we = o.search_all(data)
            cnt, first, st, en = run(p, data)
            assert (cnt == wc.astype(np.int32)).all(), p
            assert (first == np.concatenate([[0], np.cumsum(wc)[:-1]])).all(), p
            assert (st == ws).all() and (en == we).all(), p
    for p in ("ab+c", "[0-9]+\\.[0-9]+"):
        r = rr.RRegex(p)
        rep = SearchReplay(r.program(rr.PROGRAM_SEARCH_FWD), r.program(rr.PROGRAM_SEARCH_REV))
        needle = {"ab+c": b"abbbbc", "[0-9]+\\.[0-9]+": b"12345.678"}[p]
        lines = []
        for n in (0, 1, 1000, 1017, 1024, 2048, 3000, 4096, 9000):
            filler = bytes(rng.choice(b"xyz ") for _ in range(n))
            lines += [filler + needle + filler + needle + b" tail " + needle, filler, needle * 3 + filler]
        data = b"\n".join(lines)
        cnt, first, st, en = run(p, data)
        for i, ln in enumerate(lines):
            want = rep.search_all(ln)
            got = [(int(st[first[i] + j]), int(en[first[i] + j])) for j in range(int(cnt[i]))]
            assert got == want, (p, i, len(ln), got[:3], want[:3])


def test_search_modes_agree_at_scale():
    """The three modes of the stripe-wise search kernel on 256 MiB of the URL config (16384 chunks, persistent
    workgroups that take several each): (1) the first chunk of lines against the oracle's brute force; (2) a line that the
    pattern accepts as a whole has a match ending inside it; (3) the all-matches pass reproduces the first-match pass:
    count > 0 exactly where a first match exists, and its first slot holds that match."""
    import synth
    n = 256 << 20
    host = synth.corpus("url", 5, n)
    dev = torch.from_numpy(host).cuda()
    corpus = rr.Corpus(dev)
    r = rr.RRegex(U2)
    s, e = r.search_corpus(corpus)
    piece = host[:256 << 10]
    k = int((piece == 10).sum())
    ws, we = OracleRegex(U2).search_lines(piece[:int(np.nonzero(piece == 10)[0][-1]) + 1])
    assert (s[:k].cpu().numpy() == ws).all() and (e[:k].cpu().numpy() == we).all()
    acc = r.match_corpus(corpus).bool()
    starts = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), torch.nonzero(dev == 10).flatten() + 1])[:corpus.num_lines]
    ends = torch.cat([torch.nonzero(dev == 10).flatten(), torch.tensor([n], device="cuda")])[:corpus.num_lines]
    length = (ends - starts).to(torch.int32)
    assert bool((e[acc] >= 0).all()) and bool((e[acc] <= length[acc]).all())
    assert bool(((s >= 0) == (e >= 0)).all()) and bool((s[e >= 0] <= e[e >= 0]).all())
    cnt, first, st, en = r.search_all(corpus)
    assert bool(((cnt > 0) == (e >= 0)).all())
    has = cnt > 0
    assert torch.equal(st[first[has]], s[has]) and torch.equal(en[first[has]], e[has])
    assert int(cnt.sum()) == st.numel()
    f2, s2, e2 = r.search_all_fused(corpus)
    assert int(f2[-1]) == st.numel() and torch.equal(f2[:-1], first) and torch.equal(s2, st) and torch.equal(e2, en)


def test_search_dense_unanchored_matches_against_the_replay():
    """Hundreds of hits per 16 KiB chunk whose start has to be walked back to (the wave's job pool fills and is walked 64 jobs at a
    time, the rest waits; several matches inside one text word; hits met while a lane follows its last line): first match and all
    matches against the CPU replay of the two search tables, which tests/test_lowering.py pins to the oracle."""
    from program_replay import SearchReplay
    rng = np.random.default_rng(23)
    cases = [("ab+", b"ab \n", [0.45, 0.45, 0.09, 0.01], 384 << 10),          # lines of ~100 bytes, a match every few bytes
             ("b+c", b"abc\n", [0.3, 0.4, 0.25, 0.05], 256 << 10),            # 20-byte lines, walks of varying length
             ("[ab]+c", b"abcx\n", [0.4, 0.4, 0.1, 0.07, 0.03], 256 << 10)]   # long walks that stop at an x or at the previous match
    for pattern, alphabet, prob, n in cases:
        data = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n, p=prob).astype(np.uint8)
        data[-1] = 10
        r = rr.RRegex(pattern)
        rep = SearchReplay(r.program(rr.PROGRAM_SEARCH_FWD), r.program(rr.PROGRAM_SEARCH_REV))
        lines = data.tobytes()[:-1].split(b"\n")
        first_want = [rep.search(ln) for ln in lines]
        all_want = [rep.search_all(ln) for ln in lines]
        _oracle_confirms_search(pattern, lines, first=first_want, every=all_want, starts_per_match=3)
        dev = torch.from_numpy(data).cuda()
        for stripe in (1024, 0):
            corpus = rr.Corpus(dev, stripe=stripe)
            assert corpus.num_lines == len(lines)
            s, e = r.search_corpus(corpus)
            got = list(zip(s.cpu().tolist(), e.cpu().tolist()))
            assert got == first_want, (pattern, stripe, next(i for i, (g, w) in enumerate(zip(got, first_want)) if g != w))
            cnt, first, st, en = r.search_all(corpus)
            assert cnt.cpu().tolist() == [len(m) for m in all_want], (pattern, stripe)
            flat = [m for ms in all_want for m in ms]
            assert list(zip(st.cpu().tolist(), en.cpu().tolist())) == flat, (pattern, stripe)
            f2, s2, e2 = r.search_all_fused(corpus)
            assert int(f2[-1]) == len(flat) and torch.equal(f2[:-1], first) and torch.equal(s2, st) and torch.equal(e2, en), (pattern, stripe)


def test_search_with_tables_beyond_the_chunk_kernels_lds():
    """`[ab]*a[ab]{11}x`: the product table of the stripe-wise search kernel (8193 rows) does not fit a CU's LDS: the kernel's
    GLOBAL form (the stride-2 table left in HBM/L2, only the pair table in LDS; rounds 1-3: line offsets, then a lane per line) -
    first match, all matches and the one-call form against the CPU replay of the two tables, with and without bytes >= 0x80."""
    from program_replay import SearchReplay
    pattern = "[ab]*a[ab]{11}x"
    r = rr.RRegex(pattern)
    line = r.program(rr.PROGRAM_SEARCH_LINE)
    assert line is not None and int(line[0]) * int(line[1]) * 4 > 160 * 1024
    assert int(r.program(rr.PROGRAM_SEARCH_LINE2)[4]) == 2                      # layout: HBM/L2 (else this test no longer reaches that form)
    rep = SearchReplay(r.program(rr.PROGRAM_SEARCH_FWD), r.program(rr.PROGRAM_SEARCH_REV))
    rng = np.random.default_rng(31)
    data = rng.choice(np.frombuffer(b"abx\n", dtype=np.uint8), size=200_000, p=[0.45, 0.45, 0.07, 0.03]).astype(np.uint8)
    data[-1] = 10
    lines = data.tobytes()[:-1].split(b"\n")
    first_want = [rep.search(ln) for ln in lines]
    all_want = [rep.search_all(ln) for ln in lines]
    _oracle_confirms_search(pattern, lines, first=first_want, every=all_want, starts_per_match=4)
    assert sum(1 for w in first_want if w[1] >= 0) > 100
    corpus = rr.Corpus(torch.from_numpy(data).cuda())
    s, e = r.search_corpus(corpus)
    assert list(zip(s.cpu().tolist(), e.cpu().tolist())) == first_want
    cnt, first, st, en = r.search_all(corpus)
    assert cnt.cpu().tolist() == [len(m) for m in all_want]
    assert list(zip(st.cpu().tolist(), en.cpu().tolist())) == [m for ms in all_want for m in ms]
    f2, s2, e2 = r.search_all_fused(corpus)
    assert int(f2[-1]) == st.numel() and torch.equal(f2[:-1], first) and torch.equal(s2, st) and torch.equal(e2, en)
    # bytes >= 0x80 in the text (they cannot index the pair table: the kernel steps them as 0x00)
    data2 = data.copy()
    data2[rng.integers(0, data2.size - 1, 400)] = rng.integers(128, 256, 400).astype(np.uint8)
    lines2 = data2.tobytes()[:-1].split(b"\n")
    first2 = [rep.search(ln) for ln in lines2]
    corpus2 = rr.Corpus(torch.from_numpy(data2).cuda())
    s, e = r.search_corpus(corpus2)
    assert list(zip(s.cpu().tolist(), e.cpu().tolist())) == first2
    f3, s3, e3 = r.search_all_fused(corpus2)
    assert list(zip(s3.cpu().tolist(), e3.cpu().tolist())) == [m for ln in lines2 for m in rep.search_all(ln)]


def test_search_with_the_forward_table_alone():
    """RRX_OPT_SEARCH_ANCHORED 0 (also what a product table beyond 65534 rows falls back to): no hit knows its start, every match is
    walked back to - the same answers as with the product table, first match, all matches and the one-call form."""
    import synth
    for pattern, kind in ((EMAIL, "email"), (U2, "url")):
        host = synth.corpus(kind, 11, 8 << 20)
        corpus = rr.Corpus(torch.from_numpy(host).cuda())
        r, r0 = rr.RRegex(pattern), rr.RRegex(pattern)
        r0.set_search_anchored(False)
        assert int(r0.program(rr.PROGRAM_SEARCH_LINE2)[0]) < int(r.program(rr.PROGRAM_SEARCH_LINE2)[0])     # fewer rows
        s, e = r.search_corpus(corpus)
        s0, e0 = r0.search_corpus(corpus)
        assert torch.equal(s, s0) and torch.equal(e, e0), pattern
        piece = host[:64 << 10]
        k = int((piece == 10).sum())
        ws, we = OracleRegex(pattern).search_lines(piece[:int(np.nonzero(piece == 10)[0][-1]) + 1])
        assert (s0[:k].cpu().numpy() == ws).all() and (e0[:k].cpu().numpy() == we).all(), pattern
        f, a, b = r.search_all_fused(corpus)
        f0, a0, b0 = r0.search_all_fused(corpus)
        assert torch.equal(f, f0) and torch.equal(a, a0) and torch.equal(b, b0), pattern
        c1 = r0.search_all(corpus)
        assert torch.equal(c1[2], a) and torch.equal(c1[3], b), pattern


def test_search_all_fused_dense_and_long():
    """rrx_search_all where the staging does not hold a chunk's matches (every byte a match: 16384 per chunk), where a
    line runs over many chunks (its owner counts and places matches far beyond its own bytes), where offsets do not fit
    16 bits (lines beyond 64 KiB), and on a corpus without a trailing newline: equal to count + scan + fill."""
    rng = np.random.default_rng(17)
    cases = []
    cases.append(("a", np.full(3 << 20, ord("a"), dtype=np.uint8)))                     # one line, every byte a match
    d = np.full(2 << 20, ord("a"), dtype=np.uint8); d[rng.integers(0, d.size, 300)] = 10
    cases.append(("a", d))                                                               # long lines of matches
    d = rng.choice(np.frombuffer(b"ab \n", dtype=np.uint8), size=4 << 20, p=[0.45, 0.45, 0.09, 0.01]).astype(np.uint8)
    cases.append(("ab+", d))
    d = rng.choice(np.frombuffer(b"abc", dtype=np.uint8), size=1 << 20).astype(np.uint8); d[200000] = 10; d[900000] = 10
    cases.append(("a{1,300}b", d))                                                       # three lines of > 64 KiB
    for pattern, data in cases:
        dev = torch.from_numpy(data).cuda()
        corpus = rr.Corpus(dev)
        r = rr.RRegex(pattern)
        cnt, first, st, en = r.search_all(corpus)
        f2, s2, e2 = r.search_all_fused(corpus)
        assert int(f2[-1]) == st.numel(), pattern
        assert torch.equal(f2[:-1], first) and torch.equal(s2, st) and torch.equal(e2, en), pattern
