"""Small driver for profiling rrx_search_all (all matches, one call): python3 search_all_run.py <workload> <bytes> <reps>.  RRX_TREE=<dir> picks
another checkout's package (A/B against an older build)."""
import os, sys, time, json
ROOT = os.environ.get("RRX_TREE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
import roaringregex_amd as rr, synth, bench
kind, pkey, _, _ = bench.WORKLOADS[sys.argv[1]]
n = int(sys.argv[2]); reps = int(sys.argv[3])
host = synth.corpus(kind, 2, n)
dev = torch.from_numpy(host).cuda()
r = rr.RRegex(bench.patterns()[pkey])
c = rr.Corpus(dev)
f, s, e = r.search_all_fused(c)
total = int(s.numel())
f, s, e = r.search_all_fused(c, cap=total); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): f, s, e = r.search_all_fused(c, cap=total)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"workload": sys.argv[1], "bytes": n, "ms": dt * 1e3, "GBs": n / dt / 1e9, "matches": total, "lib": rr.__file__}))
