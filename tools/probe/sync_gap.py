"""Does a host synchronisation between launches change a kernel's duration?  Times the indexed match (two-pass entry,
asynchronous) and the one-shot entry (synchronous by contract), each back to back and with a synchronize per call."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np
import torch
import roaringregex_amd as rr
import bench, synth

w = sys.argv[1] if len(sys.argv) > 1 else "url"
kind, pkey, nbytes, _ = bench.WORKLOADS[w]
host = np.empty(nbytes, dtype=np.uint8)
synth.fill(kind, 2, host, threads=16)
dev = torch.from_numpy(host).cuda()
regex = rr.RRegex(bench.patterns()[pkey])
corpus = rr.Corpus(dev)
n = corpus.num_lines
out = torch.empty((n + 64 + 31) // 32 + 4, dtype=torch.int32, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(64)]

def run(label, call, sync_each, reps=24):
    for _ in range(4): call()
    torch.cuda.synchronize()
    for i in range(reps):
        ev[2 * i].record(); call(); ev[2 * i + 1].record()
        if sync_each: torch.cuda.synchronize()
    torch.cuda.synchronize()
    ms = [ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(reps)]
    print("%-34s first 4: %s   last 8 avg %.4f ms" % (label, " ".join("%.3f" % x for x in ms[:4]), sum(ms[-8:]) / 8), flush=True)

two = lambda: regex.match_corpus_bits(corpus, out=out)
one = lambda: regex.match_device_bits(dev, cap_lines=n + 64, out=out)
for rep in range(2):
    run("indexed match, back to back", two, False)
    run("indexed match, sync per call", two, True)
    run("one-shot entry (sync inside)", one, False)
small = torch.empty((n + 31) // 32 + 4, dtype=torch.int32, device="cuda")
one_alloc = lambda: regex.match_device_bits(dev, cap_lines=n + 64, out=None)
run("one-shot entry, out allocated per call", one_alloc, False)
run("one-shot entry, out preallocated", one, False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ms = []
for _ in range(12):
    e0.record()
    _, n1 = regex.match_device_bits(dev, cap_lines=n + 64, out=None)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
print("bench-style loop:", " ".join("%.3f" % x for x in ms))
