"""Phase timestamps of match_stripes2_kernel (tools/probe/stamps), one row per WAVE.  usage: run.py <workload> <bytes> [stripe]
Times in microseconds relative to the first wave's entry."""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
os.environ["RRX_LIB"] = os.path.join(HERE, "librrx_stamps.so")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
import roaringregex_amd as rr, synth, bench
workload, nbytes = sys.argv[1], int(sys.argv[2])
stripe = int(sys.argv[3]) if len(sys.argv) > 3 else 0
units = int(sys.argv[4]) if len(sys.argv) > 4 else 0
kind, pkey, _, _ = bench.WORKLOADS[workload]
pattern = bench.patterns()[pkey]
host = np.empty(nbytes, dtype=np.uint8); synth.fill(kind, 2, host, threads=8)
dev = torch.from_numpy(host).cuda()
corpus = rr.Corpus(dev, stripe=stripe)
r = rr.RRegex(pattern)
r.set_background_order(False)
if units: r.set_units_per_workgroup(units)
L = rr._L
L.rrx_probe_stamp_columns.restype = C.c_int
cols = L.rrx_probe_stamp_columns()
WAVES = 16
nwg = (corpus.num_bytes // corpus.stripe + 1023) // 1024 + 1
bits = torch.empty((corpus.num_lines + 31) // 32 + 4, dtype=torch.int32, device="cuda")
stamps = torch.zeros(nwg * WAVES * cols, dtype=torch.int64, device="cuda")
L.rrx_probe_match_stamped.argtypes = [C.c_void_p] * 6
rounds = torch.zeros(nwg * WAVES * 32, dtype=torch.int64, device="cuda")
ref = r.match_corpus_bits(corpus).clone()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(6):
    stamps.zero_(); rounds.zero_(); torch.cuda.synchronize()
    ev[0].record()
    rc = L.rrx_probe_match_stamped(r._h, corpus._h, C.c_void_p(bits.data_ptr()), C.c_void_p(stamps.data_ptr()), C.c_void_p(rounds.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    ev[1].record(); torch.cuda.synchronize()
    assert rc == 0, rr._L.rrx_last_error()
assert torch.equal(bits[:ref.numel()], ref)
ms = ev[0].elapsed_time(ev[1])
S = stamps.cpu().numpy().reshape(nwg, WAVES, cols)
S = S[S[:, 0, 0] != 0]
names = ["entry", "tables loaded", "first round done", "main loop done", "follow done", "window out"]
t0 = S[:, :, 0].min()
T = (S[:, :, :6] - t0) / 100.0            # 100 MHz -> us ; [wg][wave][phase]
np.save(os.path.join(ROOT, "gpurun_out", "stamps_%s_%d.npy" % (workload, nbytes >> 20)), S)
print("== units/wg %d" % units)
print("== workload %s  %d MiB  stripe %d  workgroups %d  event time %.1f us  stamped span %.1f us" % (workload, nbytes >> 20, corpus.stripe, len(S), ms * 1e3, T[:, :, 5].max()))
def q(a): return "min %7.1f  med %7.1f  p90 %7.1f  max %7.1f" % (a.min(), np.median(a), np.percentile(a, 90), a.max())
W = T.reshape(-1, 6)
print("-- per wave (%d waves)" % len(W))
for k, n in enumerate(names): print("   at  %-18s %s" % (n, q(W[:, k])))
for k in range(1, 6): print("   in  %-18s %s" % (names[k - 1] + " ->", q(W[:, k] - W[:, k - 1])))
print("-- per workgroup: slowest wave's follow-done minus fastest wave's (intra-workgroup skew): " + q(T[:, :, 4].max(1) - T[:, :, 4].min(1)))
print("   workgroup lifetime (entry -> window out): " + q(T[:, :, 5].max(1) - T[:, :, 0].min(1)))
print("   last follow-done of the workgroup -> window out (barrier + write-out): " + q(T[:, :, 5].max(1) - T[:, :, 4].max(1)))
end = T[:, :, 5].max(1)
xcc = S[:, 0, 6]
print("-- workgroup end time by XCC: " + "  ".join("%d: med %.0f max %.0f" % (x, np.median(end[xcc == x]), end[xcc == x].max()) for x in np.unique(xcc)))
hw = S[:, 0, 7]
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7   # HW_ID: wave 3:0 simd 5:4 pipe 7:6 cu 11:8 sh 12 se 15:13
key = xcc * 1000 + se * 100 + sh * 10 + cu
u, cnt = np.unique(key, return_counts=True)
print("-- distinct (xcc, se, sh, cu): %d; workgroups per CU: min %d max %d" % (len(u), cnt.min(), cnt.max()))
order = np.argsort(end)
print("-- ten slowest workgroups: " + " ".join("[wg %d xcc %d se %d cu %d end %.0f]" % (i, xcc[i], se[i], cu[i], end[i]) for i in order[-10:]))
print("-- ten fastest workgroups: " + " ".join("[wg %d xcc %d se %d cu %d end %.0f]" % (i, xcc[i], se[i], cu[i], end[i]) for i in order[:10]))
# does a CU's pair of workgroups end together?
pairs = [end[key == k] for k in u if (key == k).sum() == 2]
if pairs:
    d = np.array([abs(p[0] - p[1]) for p in pairs]); m = np.array([max(p) for p in pairs])
    print("-- CUs with two workgroups: |end difference| " + q(d) + " ; CU end " + q(m))

# ---- the chip's progress curve: text consumed by time t (a wave has consumed r rounds of 64 x 128 bytes when it starts round r)
Rn = rounds.cpu().numpy().reshape(nwg, WAVES, 32)[:len(S)]
nr = corpus.stripe // 128
starts = (Rn[:, :, :nr].reshape(-1, nr).astype(np.float64) - t0) / 100.0      # [wave][r] us
done = T[:, :, 3].reshape(-1)                                                 # main loop done = round nr consumed
ev_t = np.concatenate([starts[:, 1:].reshape(-1), done])                      # each event = one more round (8 KiB per wave) consumed
ev_t.sort()
span = T[:, :, 5].max()
edges = np.linspace(0, span, 26)
cnt, _ = np.histogram(ev_t, bins=edges)
print("-- chip rate over time (rounds consumed per bin x 8 KiB / bin width), TB/s:")
print("   " + " ".join("%.0f-%.0fus:%.2f" % (edges[i], edges[i + 1], cnt[i] * 8192 / ((edges[i + 1] - edges[i]) * 1e-6) / 1e12) for i in range(25)))
