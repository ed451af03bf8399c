#!/bin/bash
# rocprofv3 counters of the one-call all-matches kernel: r04_search_all_prof.sh <tag> <workload> <bytes> [tree]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
TAG=$1; W=$2; N=$3; [ -n "$4" ] && export RRX_TREE=$R/$4
OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/probe/search_all_run.py $W $N 5 > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 tools/probe/search_all_run.py $W $N 1 > $OUT/pmc_sq.log 2>&1 || { tail -5 $OUT/pmc_sq.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAVES SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 tools/probe/search_all_run.py $W $N 1 > $OUT/pmc_sq2.log 2>&1 || { tail -5 $OUT/pmc_sq2.log; exit 1; }
python3 - <<PY > $OUT/summary.txt
import csv, glob, os
out = "$OUT"
print("# rocprofv3, tools/probe/search_all_run.py $W $N tree=${4:-.} (all matches in one call, rrx_search_all)")
for f in glob.glob(os.path.join(out, 'kt', '**', '*kernel_stats.csv'), recursive=True):
    for row in list(csv.DictReader(open(f)))[:4]:
        print('%-70s calls %s avg_ns %s pct %s' % (row['Name'][:70], row['Calls'], row['AverageNs'], row['Percentage']))
for d in ('pmc_sq', 'pmc_sq2'):
    for f in glob.glob(os.path.join(out, d, '**', '*counter_collection.csv'), recursive=True):
        agg = {}
        for row in csv.DictReader(open(f)):
            if 'search' not in row['Kernel_Name']: continue
            a = agg.setdefault(row['Counter_Name'], [0, 0.0]); a[0] += 1; a[1] += float(row['Counter_Value'])
        for k, (n, v) in sorted(agg.items()): print('  %-24s n=%d mean=%.6g  per byte-lane %.3f' % (k, n, v / n, v / n / ($N / 64.0)))
PY
cat $OUT/summary.txt; grep '^{' $OUT/kt.log | cut -c1-200
