#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for tree in . .oldtree; do
  OUT=$R/gpurun_out/ic_$(basename $tree); rm -rf $OUT; mkdir -p $OUT
  RRX_TREE=$R/$tree timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -- python3 tools/probe/search_all_run.py arepeat 1073741824 1 > $OUT/log 2>&1
  python3 - <<PY
import csv, glob
agg={}
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if 'search' not in row['Kernel_Name']: continue
        a=agg.setdefault(row['Counter_Name'],[0,0.0]); a[0]+=1; a[1]+=float(row['Counter_Value'])
print("$tree", {k: round(v/n) for k,(n,v) in sorted(agg.items())})
PY
done
