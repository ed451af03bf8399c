"""Explicit items with separators (trim 1): the stride-2 items kernel (RRX_OPT_ITEMS_STRIDE2, default) against the byte-stride items
kernel on the same batches - the lines of 1 GiB synthetic corpora as items - one call (rrx_match_extents) and indexed
(rrx_match_items); results compared with the corpus path."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np, torch
import roaringregex_amd as rr
import bench, synth

def timed(f, n=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

for w in sys.argv[1:] or ("url", "email", "kwlog", "arepeat"):
    kind, pkey, _, _ = bench.WORKLOADS[w]
    n = 1 << 30
    host = synth.corpus(kind, 3, n)
    dev = torch.from_numpy(host).cuda()
    nl = torch.nonzero(dev == 10).flatten()
    off = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), nl + 1]).contiguous()
    pat = bench.patterns()[pkey]
    corpus = rr.Corpus(dev)
    r = rr.RRegex(pat)
    want = r.match_corpus(corpus)[:off.numel() - 1]
    items = rr.Items(dev, off, trim=1)
    out = torch.empty(items.num_items, dtype=torch.uint8, device="cuda")
    for label, on in (("stride-2", True), ("byte-stride", False), ("stride-2", True), ("byte-stride", False)):
        r.set_items_stride2(on)
        got = r.match_extents(dev, off, trim=1)
        assert torch.equal(got.bool(), want.bool()), (w, label, "one call")
        out.zero_(); r.match_items(items, out=out)
        assert torch.equal(out.bool(), want.bool()), (w, label, "indexed")
        t1 = timed(lambda: r.match_extents(dev, off, trim=1))
        t2 = timed(lambda: r.match_items(items, out=out))
        print("%-8s %-12s items %9d  one call %7.1f GB/s (%.3f ms)   indexed %7.1f GB/s (%.3f ms)" % (w, label, off.numel() - 1, n / t1 / 1e9, t1 * 1e3, n / t2 / 1e9, t2 * 1e3), flush=True)
    tc = timed(lambda: r.match_corpus(corpus))
    print("%-8s as a corpus: %7.1f GB/s (%.3f ms)" % (w, n / tc / 1e9, tc * 1e3), flush=True)
    del dev, nl, off, corpus, items, out
