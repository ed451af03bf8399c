#!/bin/bash
# Same-box A/B of two builds (RRX_LIB) on chosen workloads: r02_lib_ab2.sh <other.so> workload...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OTHER=$R/$1; shift
run() { timeout -k 10 200 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.path.basename(os.environ.get('RRX_LIB','librrx.so')), ' '.join(sys.argv[1:]), d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['config']['accepted_rank0'])" "$@"; }
for rep in 1 2 3; do
  for W in "$@"; do
    unset RRX_LIB; run --workload $W
    export RRX_LIB=$OTHER; run --workload $W
  done
done
