"""Times the four modes of the stripe-wise search kernel apart: modes.py <workload> <bytes>"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
import roaringregex_amd as rr, synth, bench
workload, nbytes = sys.argv[1], int(sys.argv[2])
kind, pkey, _, _ = bench.WORKLOADS[workload]
host = np.empty(nbytes, dtype=np.uint8); synth.fill(kind, 2, host, threads=16)
dev = torch.from_numpy(host).cuda()
corpus = rr.Corpus(dev)
r = rr.RRegex(bench.patterns()[pkey])
def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): out = f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out
import ctypes as C
L, n = rr._L, corpus.num_lines
count = torch.zeros(n, dtype=torch.int32, device="cuda")
def do_count():
    rr._check(L.rrx_search_all_count(r._h, corpus._h, C.c_void_p(count.data_ptr()), rr._stream_ptr(None)))
t_first, _ = timed(lambda: r.search_corpus(corpus))
t_count, _ = timed(do_count)
inclusive = torch.cumsum(count, dim=0, dtype=torch.int64)
first = (inclusive - count).contiguous()
total = int(inclusive[-1].item())
start = torch.empty(total, dtype=torch.int32, device="cuda"); end = torch.empty(total, dtype=torch.int32, device="cuda")
def do_fill():
    rr._check(L.rrx_search_all_fill(r._h, corpus._h, C.c_void_p(first.data_ptr()), C.c_void_p(start.data_ptr()), C.c_void_p(end.data_ptr()), rr._stream_ptr(None)))
t_fill, _ = timed(do_fill)
t_all, _ = timed(lambda: r.search_all_fused(corpus, cap=total))
print("%s %d MiB: first %.3f ms  count %.3f  fill %.3f  one call %.3f  (matches %d)" % (workload, nbytes >> 20, t_first, t_count, t_fill, t_all, total))
