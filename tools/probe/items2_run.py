"""Minimal driver for profiling the indexed items kernels: items2_run.py <workload> <stride2: 0|1> <iterations> [MiB]
(offsets made on the host with numpy; nothing but the index, the match kernel and expand_bits on the device)."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np, torch
import roaringregex_amd as rr
import bench, synth
w, on, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = (int(sys.argv[4]) if len(sys.argv) > 4 else 1024) << 20
kind, pkey, _, _ = bench.WORKLOADS[w]
host = synth.corpus(kind, 3, n)
off_h = np.concatenate([[0], np.nonzero(host == 10)[0] + 1]).astype(np.int64)
dev = torch.from_numpy(host).cuda()
off = torch.from_numpy(off_h).cuda()
r = rr.RRegex(bench.patterns()[pkey])
r.set_items_stride2(bool(on))
items = rr.Items(dev, off, trim=1)
out = torch.empty(items.num_items, dtype=torch.uint8, device="cuda")
for _ in range(iters):
    r.match_items(items, out=out)
torch.cuda.synchronize()
print("ok", w, on, int(out.sum().item()))
