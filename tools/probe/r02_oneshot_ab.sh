#!/bin/bash
# One-shot entry (rrx_match_device), same box: the shipped library against another build (RRX_LIB).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OTHER=$R/$1
run() { timeout -k 10 200 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); c=d['cold']; print(os.path.basename(os.environ.get('RRX_LIB','librrx.so')), ' '.join(sys.argv[1:]), 'kernel_ms', d['roofline']['kernel_ms_avg'], 'one_shot_ms', c['one_shot_ms'], 'frac', c['one_shot_frac'])" "$@"; }
for rep in 1 2 3; do
  for W in url email kwlog; do
    unset RRX_LIB; run --workload $W
    export RRX_LIB=$OTHER; run --workload $W
  done
done
