"""From a rocprofv3 --kernel-trace CSV of bench.py: the kernels of every one-shot call (rrx_match_device), their durations
and the device-side span of the call (first kernel start -> last kernel end), next to bench.py's own event time per call.
Answers VERDICT r2 #6: was the 10.6 ms call a kernel or a host gap?   usage: one_shot_calls.py <dir with the csv> [bench json]"""
import csv, glob, json, os, re, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
calls, cur = [], None
for s, e, n in rows:
    if "onepass" in n:
        cur = {"k": [], "start": s}
        calls.append(cur)
    if cur is not None:
        m = re.search(r"(\w+_kernel)", n)
        short = m.group(1) if m else n[:40]
        cur["k"].append((short, (e - s) / 1e3))
        cur["end"] = e
        if "mail_results" in n:
            cur = None
bench = None
if len(sys.argv) > 2:
    try:
        bench = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])["cold"]["one_shot_ms_each_call"]
    except Exception:
        bench = None
print("one-shot calls found:", len(calls))
for i, c in enumerate(calls):
    span = (c["end"] - c["start"]) / 1e6
    ks = " ".join("%s=%.1fus" % k for k in c["k"])
    host = (" bench_event_ms=%.3f" % bench[i]) if bench and i < len(bench) else ""
    print("call %2d: device span %.3f ms%s | %s" % (i, span, host, ks))
