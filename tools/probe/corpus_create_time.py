"""Time of rr.Corpus(dev) with the automatic stripe (line-length sample + index; re-index if the sample misled)."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np, torch
import roaringregex_amd as rr
import bench, synth
for w in ("kwlines", "url", "arepeat"):
    kind, pkey, n, _ = bench.WORKLOADS[w]
    host = np.empty(n, dtype=np.uint8); synth.fill(kind, 2, host, threads=16)
    dev = torch.from_numpy(host).cuda()
    for _ in range(2): c = rr.Corpus(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): c = rr.Corpus(dev)
    torch.cuda.synchronize()
    print("%-8s %5.1f GiB  stripe %5d  Corpus() %.2f ms" % (w, n / 2**30, c.stripe, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    del dev, c
