"""Pins the oracle's set primitives (rro_bs_*) against the reference's OWN BitSet.cc, compiled unmodified
into oracle/_ref/libref_bitset.so (the only reference TU that builds without the un-vendored CRoaring).
CPU only; skipped when neither /root/reference nor a prebuilt oracle/_ref is present."""
import ctypes as C
import random

import pytest
from pyoracle import lib, ref_bitset_lib



def _ref():
    """Loaded inside the test, not at collection: a `-m gpu` run collects this module too, and the compiled reference TU
    must not be mapped into a process that never uses it."""
    ref = ref_bitset_lib()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    return ref


def _arr(W, words):
    # BitSet<W> is alignas(W*8); the shim copies into properly aligned objects, any buffer works here
    return (C.c_uint64 * 4)(*(list(words) + [0] * (4 - W)))


def _rand_words(rng, W):
    mode = rng.randrange(4)
    if mode == 0:
        return [rng.getrandbits(64) for _ in range(W)]
    if mode == 1:
        return [rng.getrandbits(64) & rng.getrandbits(64) & rng.getrandbits(64) for _ in range(W)]
    if mode == 2:
        return [0] * W
    w = [0] * W
    b = rng.randrange(64 * W)
    w[b >> 6] = 1 << (b & 63)
    return w


@pytest.mark.parametrize("W", [1, 2, 4])
def test_set_primitives_match_reference_bitset(W):
    O = lib()
    REF = _ref()
    rng = random.Random(1000 + W)
    for f in ("ref_bs_cardinality", "ref_bs_and_cardinality", "ref_bs_iterate"):
        getattr(REF, f).restype = C.c_uint32
    for f in ("rro_bs_cardinality", "rro_bs_and_cardinality", "rro_bs_iterate"):
        getattr(O, f).restype = C.c_uint32
    for it in range(400):
        a, b = _rand_words(rng, W), _rand_words(rng, W)
        # |= and &=   (BitSet.cc:8-35)
        for name in ("or", "and"):
            x, y = _arr(W, a), _arr(W, a)
            getattr(REF, "ref_bs_" + name)(W, x, _arr(W, b))
            getattr(O, "rro_bs_" + name)(W, y, _arr(W, b))
            assert list(x)[:W] == list(y)[:W]
        # cardinality / and_cardinality (BitSet.cc:36-41, 104-109)
        assert REF.ref_bs_cardinality(W, _arr(W, a)) == O.rro_bs_cardinality(W, _arr(W, a))
        assert REF.ref_bs_and_cardinality(W, _arr(W, a), _arr(W, b)) == O.rro_bs_and_cardinality(W, _arr(W, a), _arr(W, b))
        # add / contains (BitSet.cc:98-115)
        t = rng.randrange(64 * W)
        x, y = _arr(W, a), _arr(W, a)
        REF.ref_bs_add(W, x, t)
        O.rro_bs_add(W, y, t)
        assert list(x)[:W] == list(y)[:W]
        assert bool(REF.ref_bs_contains(W, _arr(W, a), t)) == bool(O.rro_bs_contains(W, _arr(W, a), t))
        # complement (BitSet.cc:42-56)
        x, y = _arr(W, a), _arr(W, a)
        REF.ref_bs_complement(W, x)
        O.rro_bs_complement(W, y)
        assert list(x)[:W] == list(y)[:W]
        # shifted copy (BitSet.cc:116-180); rotate stays below the set width as in Parser.cpp:81
        rot = rng.randrange(1, 64 * W)
        x, y = _arr(W, [0] * W), _arr(W, [0] * W)
        REF.ref_bs_shl(W, x, _arr(W, a), rot)
        O.rro_bs_shl(W, y, _arr(W, a), rot)
        assert list(x)[:W] == list(y)[:W], (W, rot, a)
        # ascending set-bit enumeration (BitSet.cc:57-97)
        o1 = (C.c_int32 * 256)()
        o2 = (C.c_int32 * 256)()
        k1 = REF.ref_bs_iterate(W, _arr(W, a), o1, 256)
        k2 = O.rro_bs_iterate(W, _arr(W, a), o2, 256)
        assert k1 == k2 and list(o1[:k1]) == list(o2[:k2])
