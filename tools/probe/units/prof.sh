#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; O=$R/gpurun_out/units_prof; rm -rf $O; mkdir -p $O
G=$((1<<30)); M2=$((2<<20))
for v in 0 16 $((M2+16)); do
  timeout -k 10 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/a_$v -- python3 tools/probe/units/one.py arepeat $G 4096 $v > $O/a_$v.log 2>&1 || { tail -5 $O/a_$v.log; exit 1; }
  timeout -k 10 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b_$v -- python3 tools/probe/units/one.py arepeat $G 4096 $v > $O/b_$v.log 2>&1 || { tail -5 $O/b_$v.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$O/*_*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name']
            if 'match_' not in k: continue
            acc[k[:60]][row['Counter_Name']]+=float(row['Counter_Value'])
        for k,v in acc.items():
            print(d.split('/')[-2], k, {c: round(x/6) for c,x in v.items()})
PY
