#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
TAG=$1
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/probe/search_run.py url 4294967296 5 > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 tools/probe/search_run.py url 4294967296 1 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 tools/probe/search_run.py url 4294967296 1 > $OUT/pmc_sq2.log 2>&1
python3 - <<PY
import csv, glob, os
out = "$OUT"
for f in glob.glob(os.path.join(out, 'kt', '**', '*kernel_stats.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        print('%-70s calls %s avg_ns %s pct %s' % (row['Name'][:70], row['Calls'], row['AverageNs'], row['Percentage']))
for d in ('pmc_sq', 'pmc_sq2'):
    for f in glob.glob(os.path.join(out, d, '**', '*counter_collection.csv'), recursive=True):
        agg = {}
        for row in csv.DictReader(open(f)):
            if 'search' not in row['Kernel_Name']: continue
            a = agg.setdefault(row['Counter_Name'], [0, 0.0]); a[0] += 1; a[1] += float(row['Counter_Value'])
        for k, (n, v) in sorted(agg.items()): print('  %-24s n=%d mean=%.6g' % (k, n, v / n))
PY
cat $OUT/kt.log | tail -2
