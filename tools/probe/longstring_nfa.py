"""Rate of rrx_match_string on ONE long device-resident string with the NFA lane engines (chunk relations)."""
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

url = np.frombuffer(b"https://www.example.com/" + b"a/b-c_d.e" * ((1 << 28) // 9), dtype=np.uint8)
ab = np.frombuffer(b"ab", dtype=np.uint8)[np.random.default_rng(1).integers(0, 2, size=1 << 28)]
for name, pat, text in (("(a|b)*a(a|b){40} (43 positions, no DFA)", "(a|b)*a(a|b){40}", ab), ("(a|b)*abb forced NFA (4 positions)", "(a|b)*abb", ab), ("U2 forced NFA (82 positions)", U2, url)):
    r = rr.RRegex(pat, rr.ENGINE_NFA)
    for n in (1 << 20, 1 << 24, 1 << 28):
        dev = torch.from_numpy(text[:n].copy()).cuda()
        ok, g = rate(r, dev)
        print("%-44s %5d MiB  %8.3f GB/s (accept=%d)" % (name, n >> 20, g, ok), flush=True)
