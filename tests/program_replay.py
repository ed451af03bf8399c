"""Test-side interpreters of the serialised device programs (rrx_program_words; layout in DESIGN.md
"Device programs").  They restate what kernels.hip's engines do per byte, in numpy/python ints, so that the
host lowering can be checked against the oracle on the CPU.  Test infrastructure only."""
import numpy as np


class NfaReplay:
    def __init__(self, words, sparse=False):
        """sparse: the wave-resident program (kind 8): CSR arrays xoff[nbits+1], xtgt[] instead of dense rows"""
        w = [int(x) for x in words]
        self.W, self.nbits, self.n_exc, self.accepts_empty = w[0], w[1], w[2], bool(w[3])
        W, o = self.W, 4

        def big(ws):
            v = 0
            for i, x in enumerate(ws):
                v |= x << (32 * i)
            return v
        self.init, self.fin, self.chain, self.self_, self.excm, self.cgrp, self.ctgt = [big(w[o + i * W:o + (i + 1) * W]) for i in range(7)]
        o += 7 * W
        self.B = [big(w[o + c * W:o + (c + 1) * W]) for c in range(256)]
        o += 256 * W
        if sparse:
            xoff, xtgt = w[o:o + self.nbits + 1], w[o + self.nbits + 1:]
            self.X = [sum(1 << q for q in xtgt[xoff[b]:xoff[b + 1]]) for b in range(self.nbits)]
        else:
            self.X = [big(w[o + b * W:o + (b + 1) * W]) for b in range(self.nbits)]
        self.mask = (1 << (32 * W)) - 1

    def accepts(self, s):
        S = self.init
        for c in s:
            if c == 0 or c >= 0x80:
                return False
            t = ((S << 1) & self.mask & self.chain) | (S & self.self_)
            t |= ((S & self.cgrp) + self.cgrp) & self.ctgt
            e = S & self.excm
            while e:
                b = (e & -e).bit_length() - 1
                e &= e - 1
                t |= self.X[b]
            S = t & self.B[c]
        return (S & self.fin) != 0

    def match_lines(self, data):
        """The LINE-MODE step of the batch kernel (kernels.hip: LineNfaEngine): no CHAIN mask (the lowering leaves a gap
        position in front of every path head), a 1 shifted into position 0 on every byte, B['\\n'] = {position 0}, the
        verdict S & FIN taken when the byte is '\\n'.  Returns one verdict per '\\n'-delimited line."""
        assert self.init == 1
        seq = bytes(data)
        if seq and seq[-1] != 10:
            seq += b"\n"                          # end of data ends the last line
        out, S = [], self.init
        for c in seq:
            if c == 10:
                out.append(1 if S & self.fin else 0)
            t = ((S << 1) & self.mask) | 1 | (S & self.self_)
            t |= ((S & self.cgrp) + self.cgrp) & self.ctgt
            e = S & self.excm
            while e:
                b = (e & -e).bit_length() - 1
                e &= e - 1
                t |= self.X[b]
            S = t & (1 if c == 10 else self.B[c])
        return out


class DfaReplay:
    def __init__(self, words):
        w = np.asarray(words, dtype=np.int64)
        self.nstates, self.ncls, self.start, self.accepts_empty = int(w[0]), int(w[1]), int(w[2]), bool(w[3])
        self.cls = w[4:260]
        self.acc = w[260:260 + self.nstates]
        self.next = w[260 + self.nstates:].reshape(self.nstates, self.ncls)

    def accepts(self, s):
        st = self.start
        for c in s:
            if c == 0 or c >= 0x80:
                return False
            st = int(self.next[st, self.cls[c]])
        return bool(self.acc[st])


class Dfa2Replay:
    """Stride-2 line-mode table (rrx_program_words kind 5): [D, C, start, accepts_empty, pair_col[128*128], next2[D][C]];
    next2 entry = next state | lines ended in the pair << 16 | their verdicts << 24 (oldest highest)."""

    def __init__(self, words):
        w = np.asarray(words, dtype=np.int64)
        self.D, self.C, self.start = int(w[0]), int(w[1]), int(w[2])
        self.pair_col = w[4:4 + 16384]
        self.next2 = w[4 + 16384:].reshape(self.D, self.C)

    def match_lines(self, data):
        """Verdict per '\\n'-delimited line of `data` (bytes < 0x80 only), stepping two bytes at a time."""
        seq = bytearray(data)
        if not seq:
            return []
        if seq[-1] != 10:
            seq.append(10)                       # end of data ends the last line
        drop_last = len(seq) % 2 == 1
        if drop_last:
            seq.append(10)                       # padding: produces one spurious empty line
        out, st = [], self.start
        for i in range(0, len(seq), 2):
            e = int(self.next2[st, self.pair_col[seq[i] * 128 + seq[i + 1]]])
            st, lines, verdicts = e & 0xffff, (e >> 16) & 0xff, e >> 24
            for k in range(lines):
                out.append((verdicts >> (lines - 1 - k)) & 1)
        return out[:-1] if drop_last else out


class Dfa2ItemsReplay:
    """The stride-2 table of explicit items with a separator byte each (rrx_program_words kind 15): [D, C, start, accepts_empty, 129,
    pair_col[129 * 129], next2[D][C]] - codes 0 ... 127 the byte values ('\\n' ordinary), 128 END OF ITEM; entries as in Dfa2Replay."""

    def __init__(self, words):
        w = np.asarray(words, dtype=np.int64)
        self.D, self.C, self.start, self.dim = int(w[0]), int(w[1]), int(w[2]), int(w[4])
        assert self.dim == 129
        self.pair_col = w[5:5 + 129 * 129]
        self.next2 = w[5 + 129 * 129:].reshape(self.D, self.C)

    def match_items(self, items):
        """Verdict per item (bytes objects): the items laid end to end, each followed by its separator, stepped two codes at a time the
        way the kernel does - a byte >= 0x80 as 0x00, the separator as 128 whatever its value."""
        seq = []
        for it in items:
            seq.extend((c if c < 128 else 0) for c in it)
            seq.append(128)
        odd = len(seq) % 2 == 1
        if odd:
            seq.append(128)                      # padding: one spurious empty item
        out, st = [], self.start
        for i in range(0, len(seq), 2):
            e = int(self.next2[st, self.pair_col[seq[i] * 129 + seq[i + 1]]])
            st, lines, verdicts = e & 0xffff, (e >> 16) & 0xff, e >> 24
            for k in range(lines):
                out.append((verdicts >> (lines - 1 - k)) & 1)
        return out[:-1] if odd else out


class SampledReplay:
    """The sampled table (rrx_program_words kinds 12 and 13): a DFA with an ESCAPE state - [D, K, start, accepts_empty, cls[256],
    accepting[D], next[D][K], escaped[D]] - and its stride-2 form, whose line ends carry TWO result bits (accepted, escaped).
    verdict(line) -> 1 / 0 / None (the table does not know); match_lines2(data) -> the same per line through the stride-2 form."""

    def __init__(self, dfa_words, dfa2_words):
        w = np.asarray(dfa_words, dtype=np.int64)
        self.D, self.K, self.start = int(w[0]), int(w[1]), int(w[2])
        self.cls = w[4:260]
        self.acc = w[260:260 + self.D]
        self.next = w[260 + self.D:260 + self.D + self.D * self.K].reshape(self.D, self.K)
        self.esc = w[260 + self.D + self.D * self.K:260 + 2 * self.D + self.D * self.K]
        w2 = np.asarray(dfa2_words, dtype=np.int64)
        self.D2, self.C2, self.start2 = int(w2[0]), int(w2[1]), int(w2[2])
        self.pair_col = w2[4:4 + 16384]
        self.next2 = w2[4 + 16384:].reshape(self.D2, self.C2)

    def verdict(self, line):
        st = self.start
        for c in line:
            st = int(self.next[st, self.cls[c] if c < 128 else 0])
        return None if self.esc[st] else int(self.acc[st])

    def match_lines2(self, data):
        seq = bytearray(data)
        if not seq:
            return []
        if seq[-1] != 10:
            seq.append(10)
        drop_last = len(seq) % 2 == 1
        if drop_last:
            seq.append(10)
        out, st = [], self.start2
        for i in range(0, len(seq), 2):
            e = int(self.next2[st, self.pair_col[seq[i] * 128 + seq[i + 1]]])
            st, nbits, bits = e & 0xffff, (e >> 16) & 0xff, e >> 24
            assert nbits % 2 == 0
            for k in range(nbits // 2):
                pair = (bits >> (nbits - 2 - 2 * k)) & 3
                out.append(None if pair & 1 else pair >> 1)
        return out[:-1] if drop_last else out


class SearchReplay:
    """The two search tables (rrx_program_words kinds 6 and 7, DFA layout) replayed the way search_stripes_kernel runs
    them: forward until accepting = smallest match end; then backwards from there, last accepting position = smallest
    start.  No byte kills the forward table; class 0 kills the reverse one (state 0)."""

    def __init__(self, fwd_words, rev_words):
        self.f, self.r = DfaReplay(fwd_words), DfaReplay(rev_words)

    def search(self, line, p=0):
        """first match of line[p:] (offsets relative to the line), or (-1, -1)"""
        f, r = self.f, self.r
        q = f.start
        if f.acc[q]:
            return p, p
        for i in range(p, len(line)):
            q = int(f.next[q, f.cls[line[i]]])
            if f.acc[q]:
                e = i + 1
                s, st = e, r.start
                for k in range(e - 1, p - 1, -1):
                    st = int(r.next[st, r.cls[line[k]]])
                    if st == 0:
                        break
                    if r.acc[st]:
                        s = k
                return s, e
        return -1, -1

    def search_all(self, line):
        """all matches left to right: continue at the end of a match, one byte further after an empty one"""
        out, p = [], 0
        while p <= len(line):
            s, e = self.search(line, p)
            if e < 0:
                break
            out.append((s, e))
            p = e if e > s else e + 1
        return out


class SearchLineReplay:
    """The stripe-wise search kernel's forward table (rrx_program_words kind 9): [nrows, ncols, start, skip, column[256],
    entry[nrows][ncols]], entry = next row | '\\n' << 16 | hit << 17 | anchored << 18.  Replayed per line as
    search_chunks_kernel runs it: step until the hit; an anchored hit starts at the line start, any other walks the
    reverse table back (SearchReplay's second half)."""

    def __init__(self, line_words, rev_words):
        w = np.asarray(line_words, dtype=np.int64)
        self.nrows, self.ncols, self.start, self.skip = int(w[0]), int(w[1]), int(w[2]), int(w[3])
        self.col = w[4:260]
        self.T = w[260:].reshape(self.nrows, self.ncols)
        self.r = DfaReplay(rev_words)

    def search(self, line):
        row = self.start
        for i, c in enumerate(line):
            e = int(self.T[row, self.col[c]])
            row = e & 0xffff
            assert not (e >> 16) & 1, "a line holds no newline"
            if (e >> 17) & 1:
                assert row == self.skip
                end = i + 1
                if (e >> 18) & 1:
                    return 0, end
                s, st = end, self.r.start
                for k in range(end - 1, -1, -1):
                    st = int(self.r.next[st, self.r.cls[line[k]]])
                    if st == 0:
                        break
                    if self.r.acc[st]:
                        s = k
                return s, end
        return -1, -1


class SearchLine2Replay:
    """The stride-2 form of the stripe-wise search kernel's forward table (rrx_program_words kind 14): [nrows, ncols, start,
    skip, layout, pair column[128][128], first[nrows][ncols], all[nrows][ncols]], entry = next row | events << 24, events =
    flags of the first byte << 2 | of the second (1 '\\n', 2 hit, 3 hit that starts at the restart point).  Replayed as
    search_chunks_kernel steps a line: pairs aligned to even offsets of the TEXT (so a line may begin on the second byte of
    a pair: `lead` bytes of the previous line, then its '\\n', come first), bytes >= 0x80 stepped as 0x00, a last odd byte
    paired with 0x00; hits that are not anchored walk the reverse table back (to the line start: first match; to the end of
    the previous match: all matches)."""

    def __init__(self, words, rev_words):
        w = np.asarray(words, dtype=np.int64)
        self.nrows, self.ncols, self.start, self.skip, self.layout = (int(x) for x in w[:5])
        self.pair = w[5:5 + 128 * 128].reshape(128, 128)
        n = self.nrows * self.ncols
        self.first = w[5 + 128 * 128:5 + 128 * 128 + n].reshape(self.nrows, self.ncols)
        self.all = w[5 + 128 * 128 + n:5 + 128 * 128 + 2 * n].reshape(self.nrows, self.ncols)
        self.r = DfaReplay(rev_words)

    def events(self, table, text, row):
        """[(position, flags)] of the bytes of `text` stepped pair by pair from `row`"""
        out = []
        t = [c if c < 128 else 0 for c in text]
        for i in range(0, len(t), 2):
            c1, c2 = t[i], t[i + 1] if i + 1 < len(t) else 0
            e = int(table[row, self.pair[c1, c2]])
            row = e & 0xffffff
            ev = e >> 24
            out.append((i, (ev >> 2) & 3))
            if i + 1 < len(t):
                out.append((i + 1, ev & 3))
        return out

    def walk_back(self, line, lo, end):
        s, st = end, self.r.start
        for k in range(end - 1, lo - 1, -1):
            st = int(self.r.next[st, self.r.cls[line[k]]])
            if st == 0:
                break
            if self.r.acc[st]:
                s = k
        return s

    def search(self, line, lead=0):
        """first match of `line`; lead = 0 / 1: the line begins on the first / second byte of a pair"""
        text = (b"\n" if lead else b"") + bytes(line) + b"\n"
        row = self.start
        for pos, f in self.events(self.first, text, row):
            p = pos - lead
            if p < 0:
                assert f == 1
                continue
            if f == 1:
                assert p == len(line)
                return -1, -1
            if f:
                return (0 if f == 3 else self.walk_back(line, 0, p + 1)), p + 1
        raise AssertionError("no newline event")

    def search_all(self, line, lead=0):
        text = (b"\n" if lead else b"") + bytes(line) + b"\n"
        out, lb = [], 0
        for pos, f in self.events(self.all, text, self.start):
            p = pos - lead
            if p < 0:
                continue
            if f == 1:
                assert p == len(line)
                return out
            if f:
                out.append((lb if f == 3 else self.walk_back(line, lb, p + 1), p + 1))
                lb = p + 1
        raise AssertionError("no newline event")
