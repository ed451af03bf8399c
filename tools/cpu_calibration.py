#!/usr/bin/env python3
"""what the `cpu_baseline` of bench.py (the oracle = our CPU restatement of the reference's loop) is worth
against the REFERENCE's own CPU path.  The reference as a whole cannot be built here (regex.h:8 needs the un-vendored CRoaring),
so the reference side of the ratio is the survey's own measurement of the unmodified reference TUs (sandbox
Xeon @ 2.1 GHz, -Ofast -flto -mavx2, per-line API); this script measures the restatement on the same two corpora shapes
(16 MiB of synthetic email / URL lines) on this machine and prints both and their ratio.  CPU only; run here or on the GPU box."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import synth
from pyoracle import OracleRegex
import bench

SURVEY = {   # MB/s, SURVEY.md section 6 [probe]: reference TUs, -Ofast, 1 thread / 8 threads (URL: 8 processes)
    "email": {"ref_1T": 46.7, "ref_8T": 264.0, "ref_O0_1T": 11.6},
    "url": {"ref_1T": 25.6, "ref_8T": 150.0, "ref_O0_1T": 3.2},
}
pats = bench.patterns()
cores = bench.host_cores()
print("host cores used: %d" % cores)
for wl, pkey in (("email", "EMAIL"), ("url", "U2")):
    data = synth.corpus(wl, 2, 16 << 20)
    o = OracleRegex(pats[pkey])
    t0 = time.perf_counter(); acc = o.match_lines(data); t1 = time.perf_counter() - t0
    one = len(data) / t1 / 1e6
    per = (len(data) // cores) >> 20 << 20
    oracles = [OracleRegex(pats[pkey]) for _ in range(cores)]
    def work(i): oracles[i].match_lines(data[i * per:(i + 1) * per])
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; tn = time.perf_counter() - t0
    many = per * cores / tn / 1e6
    s = SURVEY[wl]
    print("%-5s restatement (oracle/rr_oracle.c -O2): %.1f MB/s on 1 core, %.1f MB/s on %d | reference (survey): %.1f MB/s 1T, %.0f MB/s 8T (-O0 as shipped: %.1f)"
          " | restatement / reference: %.2f (1 core), accepted %d of %d lines"
          % (wl, one, many, cores, s["ref_1T"], s["ref_8T"], s["ref_O0_1T"], one / s["ref_1T"], int(acc.sum()), len(acc)))
