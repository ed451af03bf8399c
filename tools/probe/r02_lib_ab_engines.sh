#!/bin/bash
# Same-box A/B of two builds (RRX_LIB) on the byte-stride table engine and the NFA lane engine.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OTHER=$R/$1; shift
run() { timeout -k 10 200 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.path.basename(os.environ.get('RRX_LIB','librrx.so')), ' '.join(sys.argv[1:]), d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'])" "$@"; }
for rep in 1 2; do
  for A in "url dfa" "kwlog dfa" "email dfa" "arepeat dfa" "url nfa" "arepeat nfa" "kwlines nfa"; do
    set -- $A
    unset RRX_LIB; run --workload $1 --engine $2
    export RRX_LIB=$OTHER; run --workload $1 --engine $2
  done
done
