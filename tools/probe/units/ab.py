"""A/B in one process: the stride-2 batch kernel with one stripe per lane and launch against the same stripes
handed out in units of 64 inside the workgroup (rrx_set_option RRX_OPT_UNITS_PER_WORKGROUP), alternating rounds on one corpus.
usage: ab.py <workload> <bytes> <stripe,units> [<stripe,units> ...]     (units 0 = the classic kernel; stripe 0 = automatic)"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
import roaringregex_amd as rr, synth, bench
workload, nbytes = sys.argv[1], int(sys.argv[2])
variants = [tuple(int(x) for x in v.split(",")) for v in sys.argv[3:]]
kind, pkey, _, _ = bench.WORKLOADS[workload]
pattern = bench.patterns()[pkey]
host = np.empty(nbytes, dtype=np.uint8); synth.fill(kind, 2, host, threads=16)
dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
for off in range(0, nbytes, 1 << 30): dev[off:off + (1 << 30)].copy_(torch.from_numpy(host[off:off + (1 << 30)]))
corpora, regexes = {}, {}
for stripe, units in variants:
    if stripe not in corpora: corpora[stripe] = rr.Corpus(dev, stripe=stripe)
    r = rr.RRegex(pattern); r.set_background_order(False); r.set_units_per_workgroup(units)
    regexes[(stripe, units)] = r
nl = next(iter(corpora.values())).num_lines
out = torch.empty((nl + 31) // 32 + 4, dtype=torch.int32, device="cuda")
ref = None
for (stripe, units), r in regexes.items():
    got = r.match_corpus_bits(corpora[stripe]).clone()
    if ref is None: ref = got
    assert torch.equal(got, ref), ("result differs", stripe, units)
    for _ in range(6): r.match_corpus_bits(corpora[stripe], out=out)
torch.cuda.synchronize()
res = {k: [] for k in regexes}
for rnd in range(4):
    for k, r in regexes.items():
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(); r.match_corpus_bits(corpora[k[0]], out=out); b.record()
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in ev]
        res[k].append(sum(ms) / len(ms))
print("# %s %d MiB, %d lines; call ms per launch (memset + kernel), four alternating rounds of 20" % (workload, nbytes >> 20, nl))
for (stripe, units), v in res.items():
    m = sum(v) / len(v)
    print("stripe %5d (%5d) units/wg %4d : %s  mean %.4f ms = %.3f of peak" % (stripe, corpora[stripe].stripe, units, " ".join("%.4f" % x for x in v), m, nbytes / m / 1e6 / 8000))
