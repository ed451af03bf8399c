"""A/B in one process: the stride-2 kernel's common flush period (RRX_OPT_FLUSH_SLOTS).  usage: flush_ab.py <workload> <bytes> <slots> [<slots> ...]   (0 = automatic)"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
import roaringregex_amd as rr, synth, bench
workload, nbytes = sys.argv[1], int(sys.argv[2])
variants = [int(x) for x in sys.argv[3:]]
kind, pkey, _, _ = bench.WORKLOADS[workload]
host = np.empty(nbytes, dtype=np.uint8); synth.fill(kind, 2, host, threads=16)
dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
for off in range(0, nbytes, 1 << 30): dev[off:off + (1 << 30)].copy_(torch.from_numpy(host[off:off + (1 << 30)]))
corpus = rr.Corpus(dev)
regexes = {}
for v in variants:
    r = rr.RRegex(bench.patterns()[pkey]); r.set_background_order(False); r.set_flush_slots(v); regexes[v] = r
out = torch.empty((corpus.num_lines + 31) // 32 + 4, dtype=torch.int32, device="cuda")
ref = None
for v, r in regexes.items():
    got = r.match_corpus_bits(corpus).clone()
    if ref is None: ref = got
    assert torch.equal(got, ref), ("result differs", v)
    for _ in range(6): r.match_corpus_bits(corpus, out=out)
torch.cuda.synchronize()
res = {k: [] for k in regexes}
for rnd in range(4):
    for k, r in regexes.items():
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(); r.match_corpus_bits(corpus, out=out); b.record()
        torch.cuda.synchronize()
        res[k].append(sum(a.elapsed_time(b) for a, b in ev) / len(ev))
print("# %s %d MiB, %d lines (mean %.1f bytes), stripe %d; call ms per launch, four alternating rounds of 20" % (workload, nbytes >> 20, corpus.num_lines, nbytes / corpus.num_lines, corpus.stripe))
for v, t in res.items():
    m = sum(t) / len(t)
    print("flush every %2d slots%s: %s  mean %.4f ms = %.3f of peak" % (v if v else -1, " (auto)" if not v else "", " ".join("%.4f" % x for x in t), m, nbytes / m / 1e6 / 8000))
