"""Rate of rrx_match_string on ONE long device-resident string: chunk-map path vs the single sequential lane."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import roaringregex_amd as rr
from patterns import U2

def rate(r, dev, reps=3):
    r.match_string(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ok = r.match_string(dev)
    torch.cuda.synchronize()
    return ok, dev.numel() * reps / (time.perf_counter() - t0) / 1e9

url = np.frombuffer(b"https://www.example.com/" + b"a/b-c_d.e" * ((1 << 30) // 9), dtype=np.uint8)
ab = np.frombuffer(b"ab", dtype=np.uint8)[np.random.default_rng(1).integers(0, 2, size=1 << 30)]
for name, pat, text in (("U2 (90 table states)", U2, url), ("(a|b)*abb (4 table states)", "(a|b)*abb", ab), (".*abc.* on a/b text", ".*abc.*", ab)):
    for n in (1 << 20, 1 << 26, 1 << 30):
        dev = torch.from_numpy(text[:n].copy()).cuda()
        ok, g = rate(rr.RRegex(pat), dev)
        line = "%-28s %5d MiB  chunk maps %8.2f GB/s (accept=%d)" % (name, n >> 20, g, ok)
        if n == 1 << 20:
            ok2, g2 = rate(rr.RRegex(pat, rr.ENGINE_NFA), dev, reps=1)
            line += "   one sequential lane (NFA engine) %.4f GB/s" % g2
        print(line, flush=True)
