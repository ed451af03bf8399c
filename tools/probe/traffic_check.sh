#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of the search kernels and of the one-shot entry on one workload: traffic_check.sh <tag> <workload> <bytes>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
TAG=$1; W=$2; N=$3
OUT=$R/gpurun_out/traffic_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 100 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/first_$c -- python3 tools/probe/search_run.py $W $N 1 > $OUT/first_$c.log 2>&1 || { tail -5 $OUT/first_$c.log; exit 1; }
  timeout -k 10 100 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/all_$c -- python3 tools/probe/search_all_run.py $W $N 1 > $OUT/all_$c.log 2>&1 || { tail -5 $OUT/all_$c.log; exit 1; }
done
python3 - <<PY
import csv, glob, os, collections
out = "$OUT"; n = $N
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(out + "/*_SIZE"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("rrx::dev::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if "rrx" in r["Kernel_Name"]: acc[os.path.basename(d).split("_")[0] + " " + k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    f = d.get("FETCH_SIZE", [0]); w = d.get("WRITE_SIZE", [0])
    fm, wm = sum(f) / len(f) * 1024 * 2, sum(w) / len(w) * 1024
    print("%-80s launches %3d  fetched %8.1f MB (%.3f x text)  written %8.1f MB" % (k, len(f), fm / 1e6, fm / n, wm / 1e6))
PY
