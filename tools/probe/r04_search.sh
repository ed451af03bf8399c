#!/bin/bash
# round-4 search measurement: r04_search.sh <tag> [tests|notests]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
TAG=${1:-r04s}; DO_TESTS=${2:-tests}
if [ "$DO_TESTS" = tests ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "search" > gpurun_out/${TAG}_tests.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests.log; exit 1; }
  tail -3 gpurun_out/${TAG}_tests.log
fi
: > gpurun_out/${TAG}_search.txt
for w in email url kwlog kwlines arepeat; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --search --steps 5 --warmup 2 >> gpurun_out/${TAG}_search.txt 2>> gpurun_out/${TAG}_search.err || { tail -5 gpurun_out/${TAG}_search.err; exit 1; }
done
python - <<PY
import json
for l in open('gpurun_out/${TAG}_search.txt'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); s=d['search']; print(d['config']['workload'][:40], 'match', d['value'], '| search first', s['GBs'], '| count+scan+fill', s['all_matches']['GBs'], '| one call', s['all_matches_one_call']['GBs'], '| lines with a match', s['lines_with_a_match'], 'matches', s['all_matches']['matches'])
PY
