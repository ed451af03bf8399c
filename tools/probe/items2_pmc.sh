#!/bin/bash
# fetched / written bytes and LDS / VALU activity of the indexed items kernels.  usage: items2_pmc.sh <tag> <workload>
tag=${1:-items2pmc}; w=${2:-url}
out=$PWD/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for on in 1 0; do
timeout -k 10 60 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/mem$on -- python3 $GRAFT_REPO_ROOT/tools/probe/items2_run.py $w $on 3 > $out/mem$on.log 2>&1 || { tail -5 $out/mem$on.log; exit 1; }
timeout -k 10 60 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $out/sq$on -- python3 $GRAFT_REPO_ROOT/tools/probe/items2_run.py $w $on 3 > $out/sq$on.log 2>&1 || { tail -5 $out/sq$on.log; exit 1; }
done
python3 - $out <<'PY'
import csv, glob, sys, collections
for sub in ("mem1", "sq1", "mem0", "sq0"):
    f = glob.glob(sys.argv[1] + "/" + sub + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("rrx::dev::(anonymous namespace)::", "").split("(")[0]
        if "item" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(sub, k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
