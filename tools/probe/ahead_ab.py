"""A/B harness of profiles/r04_ahead_bytes_ab.txt: the automatic flush word against an explicit flush period of the same length, same process, alternating.  While the
experiments of that file were in the tree the two took different paths of the stride-2 kernel (bytes requested behind the stripe; the build for very short lines); in the
shipping tree both are the same kernel, so this now shows the noise of such a comparison.  Run it from a second checkout to compare builds (tools/probe/ab_oldtree.sh).
usage: ahead_ab.py [workload ...]"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np, torch
import roaringregex_amd as rr
import bench, synth
for w in sys.argv[1:] or ("kwlines", "email", "url", "kwlog", "arepeat"):
    kind, pkey, nbytes, _ = bench.WORKLOADS[w]
    host = np.empty(nbytes, dtype=np.uint8)
    synth.fill(kind, 2, host)
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for off in range(0, nbytes, 1 << 30):
        dev[off:off + (1 << 30)].copy_(torch.from_numpy(host[off:off + (1 << 30)]))
    del host
    corpus = rr.Corpus(dev)
    avg = nbytes // corpus.num_lines
    slots = 1
    while slots < 32 and slots * 2 * 16 <= avg * 16:
        slots *= 2
    r = rr.RRegex(bench.patterns()[pkey])
    r.set_background_order(False)
    out = r.match_corpus_bits(corpus).clone()
    want = out.clone()
    res = {}
    for rnd in range(3):
        for label, fs in (("auto", 0), ("all 128 bytes", slots)):
            r.set_flush_slots(fs)
            for _ in range(3): r.match_corpus_bits(corpus, out=out)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): r.match_corpus_bits(corpus, out=out)
            b.record(); torch.cuda.synchronize()
            assert torch.equal(out, want)
            res.setdefault(label, []).append(a.elapsed_time(b) / 10)
    print("%-8s stripe %5d  mean line %4d B | " % (w, corpus.stripe, avg) + " | ".join("%s: %s ms (%.0f GB/s)" % (k, " ".join("%.4f" % x for x in v), nbytes / (sum(v) / len(v)) / 1e6) for k, v in res.items()), flush=True)
    del dev, corpus
