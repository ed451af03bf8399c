"""The C-ABI library loads on a CPU-only box and exports exactly what include/rrx.h declares."""
import ctypes
import os
import re

import roaringregex_amd as rr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_library_agree():
    hdr = open(os.path.join(ROOT, "include", "rrx.h")).read()
    declared = set(re.findall(r"\b(rrx_[a-z_]+)\s*\(", hdr))
    assert declared == set(rr.ABI_SYMBOLS)
    lib = ctypes.CDLL(os.path.join(ROOT, "roaringregex_amd", "librrx.so"))
    for name in declared:
        assert hasattr(lib, name), name


def test_compile_needs_no_device():
    r = rr.RRegex("ab*c")
    assert r.states_n == 6 and r.engine_name in ("dfa-stride2-table", "dfa-wide-table", "dfa-classed-table", "nfa-shift-and")
