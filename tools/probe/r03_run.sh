#!/bin/bash
# round-3 measurement set.  usage: r03_run.sh <tag> [tests|notests]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
TAG=${1:-r03}; DO_TESTS=${2:-tests}
if [ "$DO_TESTS" = tests ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/${TAG}_tests.log; exit 1; }
  tail -3 gpurun_out/${TAG}_tests.log
fi
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_default.json 2> gpurun_out/${TAG}_default.err || { tail -5 gpurun_out/${TAG}_default.err; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/${TAG}_default.json').read().strip().splitlines()[-1]);print(d['value'],d['roofline']['frac'],d['roofline']['kernel_ms_avg'],d['cold'])"
