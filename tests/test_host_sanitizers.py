"""The host compile pipeline (front end, reduction with the worklist quotients, NFA / DFA / stride-2 lowering, search tables, the
profiled table order) under AddressSanitizer + UBSan on ~420 patterns including the large ones.  CPU only (GPU sanitizers are not
available on this pool); the harness is tools/sanitize/host_pipeline_asan.cpp."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_pipeline_is_clean_under_asan_and_ubsan():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "tools", "sanitize")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "patterns lowered" in out.stdout and "ERROR" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-2000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_order_search_is_clean_under_tsan():
    """The background table-order search: deciders race (first match / rrx_order_table / skip), pollers
    read the state the way rrx_table_order does, the owner dies while the search may still run - ThreadSanitizer, CPU build."""
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "tools", "sanitize"), "tsan"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "no report" in out.stdout and "WARNING: ThreadSanitizer" not in out.stderr, out.stderr[-2000:]
