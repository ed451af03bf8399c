"""Indexed explicit items (rrx_match_items) at several batch sizes: is the 1 GiB rate the kernel's, or a launch's fixed cost?"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np, torch
import roaringregex_amd as rr
import bench, synth

kind, pkey, _, _ = bench.WORKLOADS["url"]
pat = bench.patterns()[pkey]
host = synth.corpus(kind, 3, 4 << 30)
for n in (1 << 30, 2 << 30, 4 << 30):
    nl = np.nonzero(host[:n] == 10)[0]                          # (torch.nonzero cannot index 2 GiB)
    last = int(nl[-1]) + 1
    dev = torch.from_numpy(host[:last]).cuda()
    off = torch.from_numpy(np.concatenate([[0], nl + 1]).astype(np.int64)).cuda()
    del nl
    r = rr.RRegex(pat)
    items = rr.Items(dev, off, trim=1)
    out = torch.empty(items.num_items, dtype=torch.uint8, device="cuda")
    r.match_items(items, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        r.match_items(items, out=out)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 5
    corpus = rr.Corpus(dev)
    r.match_corpus(corpus)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        r.match_corpus(corpus)
    torch.cuda.synchronize()
    t2 = (time.perf_counter() - t0) / 5
    print("url %5d MiB  items indexed %7.1f GB/s (%.3f ms)  stripe-wise=%s | same bytes as a corpus %7.1f GB/s (%.3f ms)" % (last >> 20, last / t / 1e9, t * 1e3, items.stripe_wise, last / t2 / 1e9, t2 * 1e3), flush=True)
    del items, out, corpus, off, dev
    torch.cuda.empty_cache()
