"""Error behaviour of the C ABI that needs no device: status codes, messages, argument validation."""
import ctypes as C
import time

import pytest

import roaringregex_amd as rr

L = rr._L


def test_status_codes_and_messages():
    h = C.c_void_p()
    assert L.rrx_compile(b"a)", C.byref(h)) == 1 and not h.value            # RRX_ERR_PATTERN
    assert b"invalid expression" in L.rrx_last_error()
    assert L.rrx_compile(None, C.byref(h)) == 2                              # RRX_ERR_ARG
    assert L.rrx_compile_ex(b"a", 99, C.byref(h)) == 2
    assert L.rrx_compile(b"ab*c", C.byref(h)) == 0 and h.value
    assert L.rrx_num_states(h) == 6 and L.rrx_set_class(h) == 1
    L.rrx_free(h)


def test_corpus_argument_validation_happens_before_any_device_call():
    c = C.c_void_p()
    assert L.rrx_corpus_create_ex(0, C.c_void_p(16), 100, 1000, None, C.byref(c)) == 2      # stripe not a power of two
    assert b"stripe" in L.rrx_last_error()
    assert L.rrx_corpus_create_ex(0, C.c_void_p(16), 100, 32768, None, C.byref(c)) == 2     # stripe too large
    assert L.rrx_corpus_create_ex(0, C.c_void_p(8), 100, 0, None, C.byref(c)) == 2          # base not 16-byte aligned
    assert b"aligned" in L.rrx_last_error()
    assert L.rrx_corpus_create(0, None, 100, None, C.byref(c)) == 2


def test_oversized_automata_are_refused_quickly():
    # beyond the front end's 65536 reference states: refused before any lowering is tried
    t0 = time.time()
    with pytest.raises(rr.RRegexError, match="too many states"):
        rr.RRegex("(a|b)*a(a|b){20000}")
    assert time.time() - t0 < 20
    # > 8192 positions and an exploding subset construction: only the wave-resident engine admits it (the group engine: 32 lanes x 8 words)
    assert rr.RRegex("(a|b)*a(a|b){5000}", rr.ENGINE_NFA_WAVE).engine_name == "nfa-group-cooperative"
    with pytest.raises(rr.RRegexError, match="too large"):
        rr.RRegex("(a|b)*a(a|b){8200}", rr.ENGINE_NFA_WAVE)
    # forcing an engine that cannot hold the automaton is refused as well
    with pytest.raises(rr.RRegexError, match="too large"):
        rr.RRegex("(a|b)*a(a|b){600}", rr.ENGINE_NFA)
    with pytest.raises(rr.RRegexError, match="too large"):
        rr.RRegex("(a|b)*a(a|b){600}", rr.ENGINE_DFA)


def test_engine_selection_ladder():
    assert rr.RRegex("abc").engine_name == "dfa-stride2-table"
    assert rr.RRegex("abc", rr.ENGINE_DFA).engine_name == "dfa-wide-table"
    assert rr.RRegex("a{1,300}", rr.ENGINE_DFA).engine_name == "dfa-classed-table"
    assert rr.RRegex("(a|b)*a(a|b){40}").engine_name == "nfa-shift-and"           # 2^41 subsets, 44 positions
    assert rr.RRegex("(a|b)*a(a|b){600}").engine_name == "nfa-group-cooperative"
    assert rr.RRegex("(a|b)*a(a|b){40}", rr.ENGINE_NFA_BLOCK).engine_name == "nfa-wave-resident"
    assert rr.RRegex("abc", rr.ENGINE_DFA_GLOBAL).engine_name == "dfa-global-table"
