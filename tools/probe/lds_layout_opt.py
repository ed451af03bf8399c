"""Can a frequency-aware layout of the stride-2 table lower its LDS bank conflicts?  (simulate first.)
T2's entry for (state s, pair column c) sits at row_base[s] + col_off[c]: the kernel reads the row's LDS address from the
previous entry and the column's byte offset from P, so BOTH are free parameters of the layout - bank = (a_s + b_c) mod 32
with a_s, b_c of our choosing (rows padded, columns permuted).  This script takes the half-waves of real corpus text
(lanes = stripes 4 KiB apart in lockstep, as lds_conflict_sim.py), measures the shipped layout (a_s = 47 s, b_c = c) and then
improves (a, b) by coordinate descent on a training sample, evaluating on a held-out sample.
usage: lds_layout_opt.py [workload] [sample MiB]"""
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
nl = stripe // 2
lanes = (mib << 20) // stripe
t = text[:lanes * stripe].reshape(lanes, stripe)
cols = pair_col[t[:, 0::2].astype(np.int64), t[:, 1::2].astype(np.int64)]
state = np.full(lanes, start, dtype=np.int64)
states = np.empty((lanes, nl), dtype=np.int64)
for s in range(nl):
    states[:, s] = state
    state = next2[state, cols[:, s]] & 0xffff
groups = lanes // 32
S = states[:groups * 32].reshape(groups, 32, nl).transpose(0, 2, 1).reshape(-1, 32)
Cc = cols[:groups * 32].reshape(groups, 32, nl).transpose(0, 2, 1).reshape(-1, 32)
rng = np.random.default_rng(1)
perm = rng.permutation(len(S))
train, test = perm[:6000], perm[6000:16000]

def prep(idx):
    s, c = S[idx], Cc[idx]
    key = s * C + c
    order = np.argsort(key, axis=1)
    ks = np.take_along_axis(key, order, 1)
    first = np.ones_like(ks, dtype=bool); first[:, 1:] = ks[:, 1:] != ks[:, :-1]      # one representative per distinct entry
    return np.take_along_axis(s, order, 1), np.take_along_axis(c, order, 1), first

def cost(a, b, data):
    s, c, first = data
    bank = (a[s] + b[c]) & 31
    onehot = (bank[:, :, None] == np.arange(32)[None, None, :]) & first[:, :, None]
    return onehot.sum(1).max(1).mean()

tr, te = prep(train), prep(test)
a0, b0 = (np.arange(D) * (C | 1)) & 31, np.arange(C) & 31
print("workload", wl, "states", D, "pair columns", C, "| distinct entries per half-wave %.1f" % tr[2].sum(1).mean())
print("shipped layout (rows of %d words)      : train %.3f  held-out %.3f cycles per half-wave" % (C | 1, cost(a0, b0, tr), cost(a0, b0, te)))
freq_s = np.bincount(S[train].ravel(), minlength=D); freq_c = np.bincount(Cc[train].ravel(), minlength=C)
print("share of lookups in the 8 hottest rows %.2f, 8 hottest columns %.2f, 32 hottest entries %.2f" % (
    np.sort(freq_s)[-8:].sum() / freq_s.sum(), np.sort(freq_c)[-8:].sum() / freq_c.sum(),
    np.sort(np.bincount((S[train] * C + Cc[train]).ravel()))[-32:].sum() / freq_s.sum()))
ORDER_ONLY = len(sys.argv) > 3 and sys.argv[3] == "order"
a, b = a0.copy(), b0.copy()
best = cost(a, b, tr)
for it in range(0 if ORDER_ONLY else 3):
    for which, arr, order in (("a", a, np.argsort(-freq_s)), ("b", b, np.argsort(-freq_c))):
        for k in order:
            if (freq_s if which == "a" else freq_c)[k] == 0:
                continue
            keep = arr[k]
            vals = []
            for v in range(32):
                arr[k] = v
                vals.append(cost(a, b, tr))
            v = int(np.argmin(vals))
            if vals[v] < best - 1e-9:
                arr[k] = v; best = vals[v]
            else:
                arr[k] = keep
    print("after pass %d: train %.3f  held-out %.3f" % (it + 1, best, cost(a, b, te)), flush=True)
print("random placement of the same entries (balls into bins): %.3f" % cost(rng.integers(0, 32, D), rng.integers(0, 32, C), te))

# ---- the same optimisation WITHOUT any corpus: half-waves sampled from walks on the automaton itself (every lane walks the
# table choosing uniformly among the pair columns that keep it alive, a line end now and then), evaluated on the real text
def walk_sample(n_half_waves, steps=64, p_newline=1.0 / 24, p_noise=0.02):
    live = [np.nonzero((next2[s] & 0xffff) != 0)[0] for s in range(D)]
    nlcols = np.nonzero(((next2[start] >> 16) & 0xff) > 0)[0]            # columns that end a line (from the start row)
    allc = np.arange(C)
    Ss, Cs = [], []
    st = np.full(32 * n_half_waves // steps + 32, start, dtype=np.int64)
    st = st[:(len(st) // 32) * 32]
    for _ in range(steps):
        col = np.empty(len(st), dtype=np.int64)
        u = rng.random(len(st))
        for i, s in enumerate(st):
            if u[i] < p_newline and len(nlcols): col[i] = rng.choice(nlcols)
            elif u[i] < p_newline + p_noise or len(live[s]) == 0: col[i] = rng.choice(allc)
            else: col[i] = rng.choice(live[s])
        Ss.append(st.reshape(-1, 32).copy()); Cs.append(col.reshape(-1, 32).copy())
        st = next2[st, col] & 0xffff
    return np.concatenate(Ss), np.concatenate(Cs)

def prep2(s, c):
    key = s * C + c
    order = np.argsort(key, axis=1)
    ks = np.take_along_axis(key, order, 1)
    first = np.ones_like(ks, dtype=bool); first[:, 1:] = ks[:, 1:] != ks[:, :-1]
    return np.take_along_axis(s, order, 1), np.take_along_axis(c, order, 1), first

ws, wc = walk_sample(4000 if not ORDER_ONLY else 64)
wtr = prep2(ws, wc)
fs = np.bincount(ws.ravel(), minlength=D); fc = np.bincount(wc.ravel(), minlength=C)
a, b = a0.copy(), b0.copy()
best = cost(a, b, wtr)
print("automaton walks: %d half-waves, distinct entries %.1f, shipped layout on them %.3f" % (len(ws), wtr[2].sum(1).mean(), best))
for it in range(0 if ORDER_ONLY else 2):
    for which, arr, order in (("a", a, np.argsort(-fs)), ("b", b, np.argsort(-fc))):
        for k in order:
            if (fs if which == "a" else fc)[k] == 0:
                continue
            keep = arr[k]
            vals = []
            for v in range(32):
                arr[k] = v
                vals.append(cost(a, b, wtr))
            v = int(np.argmin(vals))
            if vals[v] < best - 1e-9:
                arr[k] = v; best = vals[v]
            else:
                arr[k] = keep
    print("walk-trained pass %d: on walks %.3f  on the real text %.3f" % (it + 1, best, cost(a, b, te)), flush=True)

# ---- what costs no memory: the ORDER of the rows and of the columns.  Row in slot k starts at word k * P (P = C | 1, odd), so its
# bank residue is k * P mod 32; column in slot j sits at word j.  Search over both permutations by swaps (a row or column tries
# the slot of the least used row or column of every other residue), trained on `train`, evaluated on the held-out sample.
P = C | 1
row_slot = np.arange(D); col_slot = np.arange(C)                  # state 0 (dead) must stay in slot 0
def ab():
    return (row_slot * P) & 31, col_slot & 31
best = cost(*ab(), tr)
print("order search: start %.3f" % best)
for it in range(2):
    for arr, freq, fixed0 in ((row_slot, freq_s, True), (col_slot, freq_c, False)):
        mult = P if arr is row_slot else 1
        for k in np.argsort(-freq):
            if freq[k] == 0 or (fixed0 and k == 0):
                continue
            res = (arr * mult) & 31
            cands = []
            for r in range(32):
                if r == res[k]:
                    continue
                js = [j for j in np.nonzero(res == r)[0] if not (fixed0 and j == 0)]
                if js:
                    cands.append(min(js, key=lambda j: freq[j]))
            improved = None
            for j in cands:
                arr[k], arr[j] = arr[j], arr[k]
                v = cost(*ab(), tr)
                if v < best - 1e-9:
                    best = v; improved = j
                arr[k], arr[j] = arr[j], arr[k]
                if improved == j:
                    pass
            if improved is not None:
                j = improved
                arr[k], arr[j] = arr[j], arr[k]
                best = cost(*ab(), tr)
    print("order search pass %d: train %.3f  held-out %.3f" % (it + 1, best, cost(*ab(), te)), flush=True)
out = os.path.join(ROOT, "tools", "probe", "ab", "t2_order_%s.txt" % wl)
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as f:
    f.write("%d %d\n%s\n%s\n" % (D, C, " ".join(map(str, row_slot)), " ".join(map(str, col_slot))))
print("wrote", out)

# ---- what the product can afford at first match (a few tens of ms on the host): a small sample, the hottest rows and columns only
if ORDER_ONLY:
    for nsamp, top in ((1024, 24), (2048, 32), (1024, 48)):
        trs = prep(perm[:nsamp])
        row_slot = np.arange(D); col_slot = np.arange(C)
        best = cost(*ab(), trs)
        fs_ = np.bincount(S[perm[:nsamp]].ravel(), minlength=D); fc_ = np.bincount(Cc[perm[:nsamp]].ravel(), minlength=C)
        evals = 0
        for arr, freq, fixed0 in ((row_slot, fs_, True), (col_slot, fc_, False)):
            mult = P if arr is row_slot else 1
            for k in np.argsort(-freq)[:top]:
                if freq[k] == 0 or (fixed0 and k == 0):
                    continue
                res = (arr * mult) & 31
                cands = []
                for r in range(32):
                    if r == res[k]:
                        continue
                    js = [j for j in np.nonzero(res == r)[0] if not (fixed0 and j == 0)]
                    if js:
                        cands.append(min(js, key=lambda j: freq[j]))
                improved = None
                for j in cands:
                    arr[k], arr[j] = arr[j], arr[k]
                    v = cost(*ab(), trs); evals += 1
                    if v < best - 1e-9:
                        best = v; improved = j
                    arr[k], arr[j] = arr[j], arr[k]
                if improved is not None:
                    arr[k], arr[improved] = arr[improved], arr[k]
        print("affordable search: %d half-waves, %d hottest rows and columns, one pass, %d evaluations: train %.3f held-out %.3f" % (nsamp, top, evals, best, cost(*ab(), te)), flush=True)
