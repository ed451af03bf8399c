"""One variant, a few launches (for rocprofv3): one.py <workload> <bytes> <stripe> <units> [launches]"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
import roaringregex_amd as rr, synth, bench
workload, nbytes, stripe, units = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 6
kind, pkey, _, _ = bench.WORKLOADS[workload]
host = np.empty(nbytes, dtype=np.uint8); synth.fill(kind, 2, host, threads=16)
dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
for off in range(0, nbytes, 1 << 30): dev[off:off + (1 << 30)].copy_(torch.from_numpy(host[off:off + (1 << 30)]))
corpus = rr.Corpus(dev, stripe=stripe)
r = rr.RRegex(bench.patterns()[pkey]); r.set_background_order(False); r.set_units_per_workgroup(units)
out = torch.empty((corpus.num_lines + 31) // 32 + 4, dtype=torch.int32, device="cuda")
for _ in range(n): r.match_corpus_bits(corpus, out=out)
torch.cuda.synchronize()
print("done", workload, stripe, units)
