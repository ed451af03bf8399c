"""Latency of rrx_match_corpus on SMALL corpora by stripe size: a lane steps its stripe byte pair after byte pair, a chain of dependent LDS lookups -
a 2 KiB stripe takes ~55 us whatever the corpus.  usage: small_corpus_stripes.py [workload]"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
import roaringregex_amd as rr
import bench, synth
w = sys.argv[1] if len(sys.argv) > 1 else "url"
kind, pkey, _, _ = bench.WORKLOADS[w]
r = rr.RRegex(bench.patterns()[pkey])
r.set_background_order(False)
for kib in (64, 1024, 16 << 10, 128 << 10, 512 << 10):
    d = torch.from_numpy(synth.corpus(kind, 5, kib << 10)).cuda()
    line = []
    for stripe in (0, 512, 1024, 2048, 4096):
        c = rr.Corpus(d, stripe=stripe)
        out = r.match_corpus_bits(c).clone()
        torch.cuda.synchronize()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps): r.match_corpus_bits(c, out=out)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        line.append("%s%d: %.1f us" % ("auto=" if stripe == 0 else "", c.stripe, t * 1e6))
    print("%s %7d KiB | " % (w, kib) + " | ".join(line), flush=True)
