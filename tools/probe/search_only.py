"""first-match search rate only: search_only.py <workload> <bytes> [reps]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
import roaringregex_amd as rr, synth, bench
workload, nbytes = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
kind, pkey, _, _ = bench.WORKLOADS[workload]
host = np.empty(nbytes, dtype=np.uint8); synth.fill(kind, 2, host, threads=16)
dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
for off in range(0, nbytes, 1 << 30): dev[off:off + (1 << 30)].copy_(torch.from_numpy(host[off:off + (1 << 30)]))
corpus = rr.Corpus(dev)
r = rr.RRegex(bench.patterns()[pkey])
s, e = r.search_corpus(corpus); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps): s, e = r.search_corpus(corpus)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
print("%s %d MiB first match: %.3f ms per call = %.0f GB/s; lines with a match %d, checksum %d" % (workload, nbytes >> 20, ms, nbytes / ms / 1e6, int((e >= 0).sum()), int(s.sum(dtype=torch.int64) + e.sum(dtype=torch.int64))))
