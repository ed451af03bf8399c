#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; O=$R/gpurun_out/sampled_prof; rm -rf $O; mkdir -p $O
for w in urltail urlalt; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $O/$w.log 2>&1 || { tail -5 $O/$w.log; exit 1; }
  python3 - <<PY
import csv, glob, json
d=json.loads([l for l in open("$O/$w.log") if l.startswith("{")][-1]); print("$w", d["value"], d["roofline"]["frac"], d["config"]["sampled_table_states_open_transitions"], "escaped", d["config"]["sampled_table_escaped_lines_last_launch"], "of", d["config"]["lines_per_gpu"])
for f in glob.glob("$O/$w/**/*kernel_stats.csv", recursive=True):
    for row in list(csv.DictReader(open(f)))[:9]: print("   ", row["Name"][:70], row["Calls"], row["AverageNs"])
PY
done
