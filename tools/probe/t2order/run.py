"""A/B in one process: the stride-2 table as numbered against the order lds_layout_opt.py found, same corpus, alternating.
usage: run.py <workload> <bytes> <order file>"""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
os.environ["RRX_LIB"] = os.path.join(HERE, "librrx_t2order.so")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
import roaringregex_amd as rr, synth, bench
workload, nbytes, order_file = sys.argv[1], int(sys.argv[2]), sys.argv[3]
kind, pkey, _, _ = bench.WORKLOADS[workload]
pattern = bench.patterns()[pkey]
host = np.empty(nbytes, dtype=np.uint8); synth.fill(kind, 2, host, threads=16)
dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
for off in range(0, nbytes, 1 << 30): dev[off:off + (1 << 30)].copy_(torch.from_numpy(host[off:off + (1 << 30)]))
corpus = rr.Corpus(dev)
lines = open(order_file).read().split("\n")
D, Cn = map(int, lines[0].split())
rows = np.array(lines[1].split(), dtype=np.uint32); cols = np.array(lines[2].split(), dtype=np.uint32)
base, tuned = rr.RRegex(pattern), rr.RRegex(pattern)
L = rr._L
L.rrx_probe_set_t2_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
rc = L.rrx_probe_set_t2_order(tuned._h, C.c_void_p(rows.ctypes.data), len(rows), C.c_void_p(cols.ctypes.data), len(cols))
assert rc == 0, L.rrx_last_error()
out = torch.empty((corpus.num_lines + 31) // 32 + 4, dtype=torch.int32, device="cuda")
ref = base.match_corpus_bits(corpus).clone()
assert torch.equal(tuned.match_corpus_bits(corpus), ref), "the order changed the result"
for r in (base, tuned):
    for _ in range(8): r.match_corpus_bits(corpus, out=out)
torch.cuda.synchronize()
res = {"base": [], "tuned": []}
for rnd in range(4):
    for name, r in (("base", base), ("tuned", tuned)):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(); r.match_corpus_bits(corpus, out=out); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        res[name].append(sum(ms) / len(ms))
for name in res:
    m = sum(res[name]) / len(res[name])
    print("%-5s kernel ms per launch, four alternating rounds of 20: %s  mean %.4f ms = %.3f of peak" % (name, " ".join("%.4f" % x for x in res[name]), m, nbytes / m / 1e6 / 8000))
