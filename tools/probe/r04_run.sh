#!/bin/bash
# round-4 measurement set.  usage: r04_run.sh <tag> [tests|notests] [all|default]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
TAG=${1:-r04}; DO_TESTS=${2:-tests}; WHAT=${3:-all}
if [ "$DO_TESTS" = tests ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests.log; exit 1; }
  tail -3 gpurun_out/${TAG}_tests.log
fi
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_default.json 2> gpurun_out/${TAG}_default.err || { tail -5 gpurun_out/${TAG}_default.err; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/${TAG}_default.json').read().strip().splitlines()[-1]);print('url',d['value'],d['roofline']['frac'],d['roofline']['kernel_ms_avg'],d['cold'])"
if [ "$WHAT" = all ]; then
  : > gpurun_out/${TAG}_all.txt
  for w in email arepeat kwlines kwlog; do
    timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline >> gpurun_out/${TAG}_all.txt 2>> gpurun_out/${TAG}_all.err || { tail -5 gpurun_out/${TAG}_all.err; exit 1; }
  done
  python - <<PY
import json
for l in open('gpurun_out/${TAG}_all.txt'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d.get('cold'))
PY
fi
