#!/usr/bin/env python3
"""Writes tests/golden/corpus_golden.json: for every BASELINE config, a small synthetic corpus (tools/synth.c,
fixed seed/size/chunk) and the ORACLE's accept vector over it, stored as line count, accepted count and the
FNV-1a-64 of the 0/1 vector.  Fixtures are data only; regenerate with this script (needs oracle/ and tools/ built).
The oracle itself is pinned by kat.json (reference answers); these vectors extend that pin to corpus scale and
let the GPU path be checked without running the oracle."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import synth  # noqa: E402
from patterns import EMAIL, K1000, K1000_CONTAINS, U2  # noqa: E402
from pyoracle import OracleRegex  # noqa: E402

CASES = [
    # name, synth kind, seed, bytes, chunk, pattern
    ("c1_abc_single_string", "c1", 1234, 1 << 20, 0, "abc"),
    ("c2_email", "email", 1, 2 << 20, 0, EMAIL),
    ("c3_url", "url", 2, 2 << 20, 0, U2),
    ("c4_arepeat", "arepeat", 3, 48 << 10, 16 << 10, "a{1,300}"),
    ("c5_kwlines", "kwlines", 4, 48 << 10, 16 << 10, K1000),
    ("c5_kwlog", "kwlog", 4, 96 << 10, 32 << 10, K1000_CONTAINS),
]

if __name__ == "__main__":
    out = []
    for name, kind, seed, nbytes, chunk, pattern in CASES:
        data = synth.corpus(kind, seed, nbytes, chunk=chunk, threads=1)
        acc = OracleRegex(pattern).match_lines(data)
        out.append({"name": name, "kind": kind, "seed": seed, "bytes": nbytes, "chunk": chunk, "pattern": pattern,
                    "corpus_fnv1a": "%016x" % synth.fnv1a(data), "lines": int(len(acc)), "accepted": int(acc.sum()),
                    "accept_fnv1a": "%016x" % synth.fnv1a(acc)})
        print(name, out[-1]["lines"], out[-1]["accepted"], out[-1]["accept_fnv1a"])
    with open(os.path.join(HERE, "corpus_golden.json"), "w") as f:
        json.dump({"source": "oracle/rr_oracle.c over tools/synth.c corpora; see make_corpus_golden.py", "cases": out}, f, indent=1)
