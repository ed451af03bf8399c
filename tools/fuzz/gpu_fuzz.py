"""One-off heavier randomized parity run on a GPU box (not part of the default suite): random patterns x corpora with
mixed line-length distributions (empty lines, one-byte lines, lines longer than a stripe, bytes outside the domain,
missing final newline) x every engine x three stripe sizes, against the oracle.  Usage: gpu_fuzz.py [seed] [patterns] [corpus bytes]"""
import os, random, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
import roaringregex_amd as rr
from patterns import random_pattern, random_text
from pyoracle import OracleError, OracleRegex

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
npat = int(sys.argv[2]) if len(sys.argv) > 2 else 120
big = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # fixed corpus size in bytes (several workgroups at stripe 1024)
rng = random.Random(seed)
t0 = time.time()
done = checked = 0
while done < npat:
    p = random_pattern(rng)
    try:
        o = OracleRegex(p)
    except OracleError:
        continue
    if o.states_n > 400:
        continue
    engines = []
    coop = (rr.ENGINE_NFA_WAVE, rr.ENGINE_NFA_BLOCK, rr.ENGINE_NFA_SPARSE) if done % 4 == 1 else ()      # the wave-level engines: every 4th pattern
    for e in (rr.ENGINE_NFA, rr.ENGINE_DFA, rr.ENGINE_DFA2, rr.ENGINE_DFA_GLOBAL) + coop:
        try:
            engines.append(rr.RRegex(p, e))
        except rr.RRegexError:
            pass
    if not engines:
        continue
    mix = rng.choice(["short", "empty-heavy", "long", "mixed"])
    lines = []
    total = 0
    seven_bit = done % 2 == 0
    target = big if big else rng.choice([3000, 40000, 300000])
    if coop:
        target = min(target, 40000)
    while total < target:
        if mix == "short": n = rng.choice([0, 1, 2, 3, 5, 8])
        elif mix == "empty-heavy": n = rng.choice([0, 0, 0, 1, 4])
        elif mix == "long": n = rng.choice([10, 200, 1500, 5000, 20000])
        else: n = rng.choice([0, 1, 7, 30, 130, 1100, 4200])
        s = random_text(rng, "abcxk01.d", n) if n <= 30 else "".join(rng.choice("abcxk01.d") for _ in range(n))
        b = s.encode()
        r = rng.random()
        # every other corpus stays 7-bit (0x00, 0x01, 0x7f only), so that NUL bytes also reach the stride-2 kernel:
        # a byte >= 0x80 anywhere routes the whole corpus to the byte-stride kernel
        bad_bytes = [0x00, 0x01, 0x7f] if seven_bit else [0x80, 0xc3, 0xff, 0x00, 0x01, 0x7f]
        if r < 0.02 and b: b = b[: len(b) // 2] + bytes([rng.choice(bad_bytes)]) + b[len(b) // 2:]
        lines.append(b)
        total += len(b) + 1
    data = b"\n".join(lines) + (b"\n" if rng.random() < 0.7 else b"")
    arr = np.frombuffer(data, dtype=np.uint8)
    want = o.match_lines(arr)
    dev = torch.from_numpy(arr.copy()).cuda() if len(arr) else torch.empty(0, dtype=torch.uint8, device="cuda")
    for stripe in (1024, 4096, 16384):
        corpus = rr.Corpus(dev, stripe=stripe)
        assert corpus.num_lines == len(want), (p, corpus.num_lines, len(want))
        for r_ in engines:
            got = r_.match_corpus(corpus).cpu().numpy()
            bad = np.nonzero(got != want)[0]
            if bad.size:
                print("MISMATCH", repr(p), r_.engine_name, "stripe", stripe, "mix", mix, "line", int(bad[0]), "of", len(want), flush=True)
                sys.exit(1)
            checked += 1
    # the one-shot entry (no index: one pass over the text with the lane engines), every engine
    for r_ in engines:
        bits, nlines = r_.match_device_bits(dev)
        assert nlines == len(want), (p, r_.engine_name, nlines, len(want))
        w = bits.cpu().numpy().view(np.uint32)
        got = ((w[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.uint8).reshape(-1)[:nlines]
        bad = np.nonzero(got != want)[0]
        if bad.size:
            print("MISMATCH one-shot", repr(p), r_.engine_name, "mix", mix, "line", int(bad[0]), "of", len(want), flush=True)
            sys.exit(1)
        checked += 1
    # search: first match per line against the oracle's brute force (short lines only: it is cubic)
    if mix in ("short", "empty-heavy") and len(arr) and len(arr) <= 50000:
        try:
            gs, ge = engines[0].search_corpus(rr.Corpus(dev))
        except rr.RRegexError as err:
            assert "too large" in str(err)
            gs = None
        if gs is not None:
            st, en = o.search_lines(data)
            gs, ge = gs.cpu().numpy(), ge.cpu().numpy()
            bad = np.nonzero((gs != st) | (ge != en))[0]
            if bad.size:
                print("MISMATCH search", repr(p), "mix", mix, "line", int(bad[0]), (int(gs[bad[0]]), int(ge[bad[0]])), "want", (int(st[bad[0]]), int(en[bad[0]])), flush=True)
                sys.exit(1)
            checked += 1
    # all matches: the one-launch entry (rrx_search_all, look-back over the chunks) against count + prefix sum + fill, on
    # every corpus mix (long lines cross many chunks; cap 1 forces the second call)
    if len(arr):
        try:
            corpus = rr.Corpus(dev)
            cnt, first, ms, me = engines[0].search_all(corpus)
        except rr.RRegexError as err:
            assert "too large" in str(err)
            cnt = None
        if cnt is not None:
            for cap in (None, 1):
                f2, s2, e2 = engines[0].search_all_fused(corpus, cap=cap)
                ok = int(f2[-1]) == ms.numel() and torch.equal(f2[:-1], first) and torch.equal(s2, ms) and torch.equal(e2, me)
                if not ok:
                    print("MISMATCH search_all", repr(p), "mix", mix, "cap", cap, "matches", int(ms.numel()), int(f2[-1]), flush=True)
                    sys.exit(1)
                checked += 1
            # ... and against the oracle's brute force where the lines are short (it is cubic in the line length)
            if mix in ("short", "empty-heavy") and len(arr) <= 50000:
                wc, ws, we = o.search_all(data)
                ok = np.array_equal(cnt.cpu().numpy().astype(np.uint32), wc) and np.array_equal(ms.cpu().numpy().astype(np.int32), ws) \
                    and np.array_equal(me.cpu().numpy().astype(np.int32), we)
                if not ok:
                    print("MISMATCH search_all against the oracle", repr(p), "mix", mix, "matches", int(ms.numel()), len(ws), flush=True)
                    sys.exit(1)
                checked += 1
    # explicit items: the lines as (offset, length) items, with their '\n' as the separator (trim 1) and squeezed together
    # (trim 0, empty lines dropped): the one-call entry (lane per item at these sizes) and an indexed batch (rrx_items: always the
    # stripe-wise kernel when the batch admits it)
    if len(arr) and arr[-1] == 10:
        nlpos = np.nonzero(arr == 10)[0]
        off1 = torch.from_numpy(np.concatenate([[0], nlpos + 1]).astype(np.int64)).cuda()
        lens = np.diff(np.concatenate([[0], nlpos + 1])) - 1
        nz = lens > 0
        d0 = torch.from_numpy(arr[arr != 10].copy()).cuda()
        off0 = torch.from_numpy(np.concatenate([[0], np.cumsum(lens[nz])]).astype(np.int64)).cuda()
        for r_ in engines[:2]:
            for label, (dd, oo, tr, ww) in (("trim 1", (dev, off1, 1, want)), ("trim 0", (d0, off0, 0, want[nz]))):
                if oo.numel() < 2 or dd.numel() == 0:
                    continue
                got = r_.match_extents(dd, oo, trim=tr).cpu().numpy()
                it = rr.Items(dd, oo, trim=tr)
                got2 = r_.match_items(it).cpu().numpy()
                if (got != ww).any() or (got2 != ww).any():
                    print("MISMATCH items", repr(p), r_.engine_name, label, "mix", mix, "first bad", int(np.nonzero(got != ww)[0][:1].sum()), flush=True)
                    sys.exit(1)
                checked += 1
        # every fourth round: the same ragged items TILED until the batch reaches the one-call entry's stripe-wise path (>= 65536
        # items and >= 8 MiB: item_index_kernel in resolve mode, the predicated fallback, the trim-0 insert) - ADVICE r3
        if done % 4 == 1 and len(nlpos) >= 8:
            reps = max(-(-70000 // len(nlpos)), -(-(9 << 20) // len(arr)))
            if reps * len(arr) <= (96 << 20):
                big1 = torch.from_numpy(np.tile(arr, reps)).cuda()
                o1 = np.concatenate([[0], nlpos + 1]).astype(np.int64)
                boff1 = torch.from_numpy(np.concatenate([[0]] + [o1[1:] + k * len(arr) for k in range(reps)]).astype(np.int64)).cuda()
                a0 = arr[arr != 10]
                o0 = np.concatenate([[0], np.cumsum(lens[nz])]).astype(np.int64)
                big0 = torch.from_numpy(np.tile(a0, reps)).cuda() if len(a0) else None
                boff0 = torch.from_numpy(np.concatenate([[0]] + [o0[1:] + k * len(a0) for k in range(reps)]).astype(np.int64)).cuda()
                for r_ in engines[:2]:
                    for label, (dd, oo, tr, ww) in (("trim 1", (big1, boff1, 1, np.tile(want, reps))), ("trim 0", (big0, boff0, 0, np.tile(want[nz], reps)))):
                        if dd is None or oo.numel() < 65537 or dd.numel() < (8 << 20):
                            continue
                        got = r_.match_extents(dd, oo, trim=tr).cpu().numpy()
                        if (got != ww).any():
                            print("MISMATCH tiled items", repr(p), r_.engine_name, label, "mix", mix, "reps", reps, "first bad", int(np.nonzero(got != ww)[0][0]), flush=True)
                            sys.exit(1)
                        checked += 1
                del big1, big0
    # ONE long string (rrx_match_string: chunk maps by convergence on the table engine, chunk relations on the NFA engine)
    # against the oracle: random text, and an accepted line repeated (accepted as a whole by starred patterns)
    if done % 3 == 0:
        texts = [np.frombuffer("".join(rng.choice("abcxk01.d") for _ in range(rng.choice([33000, 70001, 150000]))).encode(), dtype=np.uint8)]
        hits = [ln for ln, w in zip(data.split(b"\n"), want) if w and 0 < len(ln) < 64]
        if hits:
            texts.append(np.frombuffer(hits[0] * (40000 // len(hits[0]) + 1), dtype=np.uint8))
        for t in texts:
            wl = o.accepts(t.tobytes())
            td = torch.from_numpy(t.copy()).cuda()
            for r_ in engines:
                if r_.engine_name.startswith("nfa-group"):
                    continue                               # (a lane group per string: minutes on long strings; the wave engines do run)
                if r_.match_string(td) != wl:
                    print("MISMATCH long string", repr(p), r_.engine_name, len(t), "want", wl, flush=True)
                    sys.exit(1)
                checked += 1
    done += 1
    if done % 20 == 0:
        print("patterns", done, "checks", checked, "%.0f s" % (time.time() - t0), flush=True)
print("fuzz ok: seed", seed, "patterns", done, "engine x stripe checks", checked)
