"""Latency of the per-string facade (rrx_match_cstr: host string in, verdict out) for short strings."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import roaringregex_amd as rr
from patterns import U2
for pat, s in (("abc", b"abc"), (U2, b"https://www.example.com/a/b?c=d#e"), ("(a|b)*abb", b"ab" * 500 + b"abb"), ("(a|b)*abb", b"ab" * 10000 + b"abb")):
    r = rr.RRegex(pat)
    for _ in range(20):
        r.get_acceptance_iter(s).advance()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        m = r.get_acceptance_iter(s).advance().value()
    t = (time.perf_counter() - t0) / n
    print("%-12s %6d bytes  %7.1f us per string  accepted=%s" % (pat[:12], len(s), t * 1e6, m is not None), flush=True)
