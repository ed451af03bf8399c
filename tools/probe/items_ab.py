"""Indexed explicit items (rrx_match_items), kernel-level A/B helper: rate on url / email items, trim 1 and trim 0."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np, torch
import roaringregex_amd as rr
import bench, synth
tag = os.path.basename(os.environ.get("RRX_LIB", "librrx.so"))
for w in ("url", "email"):
    kind, pkey, _, _ = bench.WORKLOADS[w]
    n = 1 << 30
    host = synth.corpus(kind, 3, n)
    dev = torch.from_numpy(host).cuda()
    nl = torch.nonzero(dev == 10).flatten()
    off = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), nl + 1]).contiguous()
    lens = np.diff(np.concatenate([[0], np.nonzero(host == 10)[0] + 1])) - 1
    nz = lens > 0
    off0 = torch.from_numpy(np.concatenate([[0], np.cumsum(lens[nz])]).astype(np.int64)).cuda()
    d0 = torch.from_numpy(host[host != 10].copy()).cuda()
    r = rr.RRegex(bench.patterns()[pkey])
    for label, (dd, oo, tr) in (("trim 1", (dev, off, 1)), ("trim 0", (d0, off0, 0))):
        items = rr.Items(dd, oo, trim=tr)
        out = torch.empty(items.num_items, dtype=torch.uint8, device="cuda")
        for _ in range(3): r.match_items(items, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): r.match_items(items, out=out)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
        print("%-16s %-6s %s  %8.1f GB/s  (%.3f ms)  accepted %d" % (tag, w, label, dd.numel() / t / 1e9, t * 1e3, int(out.sum())), flush=True)
