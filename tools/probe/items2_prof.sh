#!/bin/bash
# kernel trace of tools/probe/items2_ab.py on one workload.  usage: items2_prof.sh <tag> <workload>
tag=${1:-items2}; w=${2:-url}
out=$PWD/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o items -- python3 $GRAFT_REPO_ROOT/tools/probe/items2_ab.py $w > $out/run.log 2>&1
tail -8 $out/run.log
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls %5s  avg %10.1f us  total %6.2f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
