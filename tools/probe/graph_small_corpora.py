"""Launch-bound use: many small corpora (or many patterns over one) - rrx_match_corpus captured in a hipGraph against the eager loop.
usage: graph_small_corpora.py [corpora] [KiB each]"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
import roaringregex_amd as rr
import bench, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
kib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
r = rr.RRegex(bench.patterns()["U2"])
r.set_background_order(False)
devs = [torch.from_numpy(synth.corpus("url", 100 + i, kib << 10)).cuda() for i in range(n)]
corpora = [rr.Corpus(d) for d in devs]
outs = [r.match_corpus_bits(c).clone() for c in corpora]
torch.cuda.synchronize()
want = [o.clone() for o in outs]

def eager():
    for c, o in zip(corpora, outs):
        r.match_corpus_bits(c, out=o)

def timed(f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

te = timed(eager)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=torch.cuda.Stream()):
    eager()
for o in outs: o.fill_(-1)
tg = timed(g.replay)
assert all(torch.equal(o, w) for o, w in zip(outs, want))
# the same launches spread over eight streams: small corpora fill a few CUs each, launches of different streams run side by side
streams = [torch.cuda.Stream() for _ in range(8)]
def spread():
    for i, (c, o) in enumerate(zip(corpora, outs)):
        with torch.cuda.stream(streams[i % 8]):
            r.match_corpus_bits(c, out=o)
for o in outs: o.fill_(-1)
ts = timed(spread)
assert all(torch.equal(o, w) for o, w in zip(outs, want))
# ... and captured with the fork / join that implies (eight parallel branches in the graph)
g8 = torch.cuda.CUDAGraph()
main = torch.cuda.Stream()
with torch.cuda.graph(g8, stream=main):
    for st in streams: st.wait_stream(main)
    spread()
    for st in streams: main.wait_stream(st)
for o in outs: o.fill_(-1)
tg8 = timed(g8.replay)
assert all(torch.equal(o, w) for o, w in zip(outs, want))
tot = n * (kib << 10)
print("  eight streams: eager %.1f us per corpus (%.1f GB/s), hipGraph with eight branches %.1f us per corpus (%.1f GB/s)"
      % (ts / n * 1e6, tot / ts / 1e9, tg8 / n * 1e6, tot / tg8 / 1e9))
print("%d corpora x %d KiB, URL regex: eager loop %.1f us per corpus (%.1f GB/s), hipGraph replay %.1f us per corpus (%.1f GB/s)"
      % (n, kib, te / n * 1e6, tot / te / 1e9, tg / n * 1e6, tot / tg / 1e9))
