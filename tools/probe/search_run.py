"""Small driver for profiling rrx_search_corpus: python3 search_run.py <workload> <bytes> <reps>"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import json
import numpy as np, torch
import roaringregex_amd as rr, synth
sys.path.insert(0, ROOT)
import bench
kind, pkey, _, _ = bench.WORKLOADS[sys.argv[1]]
n = int(sys.argv[2]); reps = int(sys.argv[3])
host = synth.corpus(kind, 2, n)
dev = torch.from_numpy(host).cuda()
r = rr.RRegex(bench.patterns()[pkey])
c = rr.Corpus(dev)
s, e = r.search_corpus(c); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): s, e = r.search_corpus(c)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"workload": sys.argv[1], "bytes": n, "ms": dt * 1e3, "GBs": n / dt / 1e9, "matches": int((e >= 0).sum().item())}))
