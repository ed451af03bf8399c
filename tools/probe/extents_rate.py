"""Rate of rrx_match_extents on a large batch of explicit items (offset array, no delimiters needed): the lines of a
synthetic corpus as items, every engine that takes them; against the batch kernel on the same bytes."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np, torch
import roaringregex_amd as rr
import bench, synth

for w in ("url", "email"):
    kind, pkey, _, _ = bench.WORKLOADS[w]
    n = 1 << 30
    host = synth.corpus(kind, 3, n)
    dev = torch.from_numpy(host).cuda()
    nl = torch.nonzero(dev == 10).flatten()
    off = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), nl + 1]).contiguous()     # item i = [off[i], off[i+1] - 1)
    pat = bench.patterns()[pkey]
    corpus = rr.Corpus(dev)
    for e in (rr.ENGINE_AUTO, rr.ENGINE_DFA, rr.ENGINE_NFA):
        r = rr.RRegex(pat, e)
        want = r.match_corpus(corpus)[:off.numel() - 1]
        got = r.match_extents(dev, off, trim=1)
        assert torch.equal(got.bool(), want.bool()), (w, r.engine_name)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r.match_extents(dev, off, trim=1)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 3
        print("%-6s %-20s items %9d  extents %8.1f GB/s  (%.2f ms)" % (w, r.engine_name, off.numel() - 1, n / t / 1e9, t * 1e3), flush=True)
    # the same items without any separator (trim 0: an Arrow-style column), empty items dropped
    keep = host != 10
    lens = np.diff(np.concatenate([[0], np.nonzero(host == 10)[0] + 1])) - 1
    nz = lens > 0
    off0 = torch.from_numpy(np.concatenate([[0], np.cumsum(lens[nz])]).astype(np.int64)).cuda()
    d0 = torch.from_numpy(host[keep].copy()).cuda()
    r = rr.RRegex(pat)
    want0 = r.match_corpus(corpus)[:off.numel() - 1][torch.from_numpy(nz).cuda()]
    got = r.match_extents(d0, off0, trim=0)
    assert torch.equal(got.bool(), want0.bool()), (w, "trim 0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r.match_extents(d0, off0, trim=0)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 3
    print("%-6s %-20s items %9d  extents %8.1f GB/s  (%.2f ms)  trim 0" % (w, r.engine_name, off0.numel() - 1, d0.numel() / t / 1e9, t * 1e3), flush=True)
    # indexed once (rrx_items), then matched: what a second, third ... pattern over the same column costs
    for label, (dd, oo, tr) in (("trim 1", (dev, off, 1)), ("trim 0", (d0, off0, 0))):
        items = rr.Items(dd, oo, trim=tr)
        out = torch.empty(items.num_items, dtype=torch.uint8, device="cuda")
        r.match_items(items, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            r.match_items(items, out=out)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 5
        print("%-6s %-20s items %9d  indexed  %8.1f GB/s  (%.2f ms)  %s  stripe-wise=%s" % (w, r.engine_name, items.num_items, dd.numel() / t / 1e9, t * 1e3, label, items.stripe_wise), flush=True)
