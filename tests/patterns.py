"""Shared pattern / input generators for the parity tests (seeded, deterministic)."""
import json
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
    KAT = json.load(f)

EMAIL = r"[A-Za-z0-9._]+@[A-Za-z0-9.]+"
U2 = [k["pattern"] for k in KAT["kat"] if k["pattern"].startswith("(http|https|ftp)")][0]
K1000 = KAT["big_states"][-1]["pattern"]
K1000_CONTAINS = ".*(" + K1000 + ").*"

ATOMS = ["a", "b", "c", "x", "[ab]", "[a-c]", "[^a]", ".", "[b-d]", "\\.", "k", "1", "0", "[0-9]"]


def random_pattern(rng, depth=0):
    """Random pattern over the reference dialect (Parser.cpp:87-150): literals, classes, groups, | * + ? {m,n}."""
    n = rng.randint(1, 4 if depth else 5)
    parts = []
    for _ in range(n):
        r = rng.random()
        if depth < 2 and r < 0.25:
            alts = [random_pattern(rng, depth + 1) for _ in range(rng.randint(1, 3))]
            atom = "(" + "|".join(alts) + ")"
        else:
            atom = rng.choice(ATOMS)
        q = rng.random()
        if q < 0.12:
            atom += "*"
        elif q < 0.22:
            atom += "+"
        elif q < 0.32:
            atom += "?"
        elif q < 0.40:
            m = rng.randint(1, 4)
            atom += "{%d}" % m
        elif q < 0.50:
            m = rng.randint(0, 3)
            atom += "{%d,%d}" % (m, m + rng.randint(1, 5))
        elif q < 0.54:
            atom += "{%d,}" % rng.randint(1, 3)
        parts.append(atom)
    return "".join(parts)


def random_text(rng, alphabet="abcxk01.d", maxlen=12):
    return "".join(rng.choice(alphabet) for _ in range(rng.randint(0, maxlen)))


def strings_near(rng, oracle, alphabet="abcxk01.d", tries=40, maxlen=14):
    """Random strings plus mutations of accepted ones, so both outcomes are exercised."""
    out = [""]
    acc = []
    for _ in range(tries):
        t = random_text(rng, alphabet, maxlen)
        out.append(t)
        if oracle.accepts(t):
            acc.append(t)
    for t in acc[:10]:
        if t:
            i = rng.randrange(len(t))
            out.append(t[:i] + rng.choice(alphabet) + t[i + 1:])
            out.append(t[:i] + t[i + 1:])
        out.append(t + rng.choice(alphabet))
    return out
