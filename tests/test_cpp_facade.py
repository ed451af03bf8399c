"""The reference's C++ interface (include/rregex.hpp) rebuilt on the C ABI: a caller shaped like the reference's
own driver (src/test/main.cpp) compiles against it with plain g++ and, on the GPU box, gives the oracle's answers."""
import os
import subprocess

import pytest

from patterns import KAT
from pyoracle import OracleRegex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "facade_driver")


def build_driver():
    src = DRIVER + ".cpp"
    lib = os.path.join(ROOT, "roaringregex_amd")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", DRIVER, src,
           "-L", lib, "-lrrx", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_facade_compiles_with_plain_gxx_and_reports_pattern_errors():
    build_driver()
    # pattern errors need no device: the reference throws std::runtime_error (Parser.cpp:36,155)
    p = subprocess.run([DRIVER], input=b"abc\n[\n", stdout=subprocess.PIPE)
    assert p.returncode == 1 and b"invalid expression!" in p.stdout


@pytest.mark.gpu
def test_facade_driver_matches_oracle_on_kat():
    if not os.path.exists(DRIVER):
        build_driver()
    for k in KAT["kat"][:16] + KAT["kat"][26:31]:
        texts = [t for t in k["accepts"] + k["rejects"] if "\n" not in t]
        if not texts or "\n" in k["pattern"]:
            continue
        o = OracleRegex(k["pattern"])
        inp = (texts[0] + "\n" + k["pattern"] + "\n" + "".join(t + "\n" for t in texts[1:])).encode("latin-1")
        p = subprocess.run([DRIVER], input=inp, stdout=subprocess.PIPE, timeout=120)
        assert p.returncode == 0, p.stdout
        lines = p.stdout.decode().strip().split("\n")
        nullable = o.accepts("")
        for t, line in zip(texts, lines):
            want = int(o.accepts(t))
            assert line.startswith("is match? %d nullable %d again %d" % (want, int(nullable), want)), (k["pattern"], t, line)
            if want:
                assert line.endswith("len %d" % len(t))
        assert lines[len(texts)] == "batch " + " ".join(str(int(o.accepts(t))) for t in texts) or (
            lines[len(texts)] == "batch" and not texts)
