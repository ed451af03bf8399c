#!/bin/bash
# bench.py --search, the shipped library against another build (RRX_LIB), same box.  usage: r02_search_lib_ab.sh <other.so> workloads...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OTHER=$R/$1; shift
run() { timeout -k 10 280 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --search --workload $1 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); s=d['search']; print(os.path.basename(os.environ.get('RRX_LIB','librrx.so')), sys.argv[1], 'first', s['GBs'], 'all', s['all_matches']['GBs'], 'one-call', s['all_matches_one_call']['GBs'])" $1; }
for W in "$@"; do
  unset RRX_LIB; run $W
  export RRX_LIB=$OTHER; run $W
done
