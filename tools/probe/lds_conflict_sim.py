"""CPU simulation of the LDS bank conflicts of the stride-2 kernel's two gathers on real corpus text, for candidate table
layouts.  Lanes of a 32-lane group = stripes 4 KiB apart, stepped in lockstep (as the kernel does).  For every pair step
the cost of a gather = max over banks of the number of DISTINCT dwords the group touches in that bank (1 = conflict-free).
usage: lds_conflict_sim.py [workload] [sample MiB]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
import roaringregex_amd as rr, synth, bench

wl = sys.argv[1] if len(sys.argv) > 1 else "url"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kind, pkey, _, _ = bench.WORKLOADS[wl]
pat = bench.patterns()[pkey]
w = rr.RRegex(pat).program(rr.ENGINE_DFA2).astype(np.int64)
D, C, start = int(w[0]), int(w[1]), int(w[2])
pair_col = w[4:4 + 16384].reshape(128, 128)
next2 = w[4 + 16384:].reshape(D, C)
stripe = 4096
text = synth.corpus(kind, 2, mib << 20)
nl = stripe // 2                                   # pair steps per stripe
lanes = (mib << 20) // stripe
t = text[:lanes * stripe].reshape(lanes, stripe)
c1, c2 = t[:, 0::2].astype(np.int64), t[:, 1::2].astype(np.int64)
cols = pair_col[c1, c2]                            # [lanes][steps]
# states per lane per step (state BEFORE the step): sequential over steps, vectorised over lanes
state = np.full(lanes, start, dtype=np.int64)
states = np.empty((lanes, nl), dtype=np.int64)
for s in range(nl):
    states[:, s] = state
    state = next2[state, cols[:, s]] & 0xffff
groups = lanes // 32

def cost(dword_addr, banks=32):
    """dword_addr [lanes][steps] -> mean over (group, step) of max distinct dwords per bank"""
    a = dword_addr[:groups * 32].reshape(groups, 32, nl).transpose(0, 2, 1).reshape(-1, 32)      # [group*step][32 lanes]
    a = np.sort(a, axis=1)
    distinct = np.ones_like(a, dtype=bool); distinct[:, 1:] = a[:, 1:] != a[:, :-1]
    bank = a % banks
    worst = np.zeros(len(a), dtype=np.int64)
    for b in range(banks):
        worst = np.maximum(worst, ((bank == b) & distinct).sum(axis=1))
    return worst.mean()

lane_id = np.arange(lanes)[:, None] % 32
print("workload", wl, "states", D, "pair columns", C, "lanes", lanes)
# P: u16 [128][130] -> dword = (c1*130 + c2) // 2
print("P  u16 [128][130]           : %.2f cycles per half-wave" % cost((c1 * 130 + c2) // 2))
print("P  u8  [128][128]           : %.2f" % cost((c1 * 128 + c2) // 4))
s2 = C | 1
for R in (1, 2, 4, 8):
    print("T2 b32 rows of %d, R=%d (%.1f KiB) : %.2f" % (s2, R, D * s2 * 4 * R / 1024, cost((states * s2 + cols) * R + lane_id % R)))
s2h = (C + 1) // 2 * 2 + 2          # u16 entries, row = s2h halves (even), dword = (state*s2h + col)//2
for R in (1, 2, 4, 8):
    print("T2 u16 rows of %d, R=%d (%.1f KiB) : %.2f" % (s2h, R, D * s2h * 2 * R / 1024, cost(((states * s2h + cols) // 2) * R + lane_id % R)))

# ---- exploration: P layouts
def distinct_mean(dword_addr):
    a = dword_addr[:groups * 32].reshape(groups, 32, nl).transpose(0, 2, 1).reshape(-1, 32)
    a = np.sort(a, axis=1)
    return (1 + (a[:, 1:] != a[:, :-1]).sum(axis=1)).mean()
print("distinct P dwords per half-wave (u16 pairs): %.1f ; distinct T2 entries: %.1f" % (distinct_mean((c1 * 130 + c2) // 2), distinct_mean(states * s2 + cols)))
for S in (129, 130, 131, 134, 138, 146, 162, 194):
    print("P u16 row stride %d halves: %.2f" % (S, cost((c1 * S + c2) // 2)))
print("P u16 [c2][c1] stride 130: %.2f" % cost((c2 * 130 + c1) // 2))
print("P u32 [128][129] (one entry per dword, 64.5 KiB): %.2f" % cost(c1 * 129 + c2))
cls = np.asarray(rr.RRegex(pat).program(rr.ENGINE_DFA)[4:260], dtype=np.int64)
K = int(cls.max()) + 2
k1 = np.where(c1 == 10, K - 1, cls[c1]); k2 = np.where(c2 == 10, K - 1, cls[c2])
print("class-pair table u32 [%d][%d]: %.2f ; distinct %.1f" % (K, K, cost(k1 * K + k2), distinct_mean(k1 * K + k2)))
print("class-pair table u16: %.2f" % cost((k1 * K + k2) // 2))
