#!/bin/bash
# Quick A/B of the bench configurations that matter (kernel ms, fraction of peak).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { timeout -k 10 200 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' '.join(sys.argv[1:]), d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['config'].get('engine'))" "$@"; }
run --workload email
run --workload arepeat
run --workload url
run --workload kwlines
run --workload kwlog
run --workload email --engine nfa
run --workload kwlog --engine nfa
run --workload email --engine dfa
run --workload url --engine dfa
run --workload email --bytes 268435456
run --workload email --bytes 536870912
