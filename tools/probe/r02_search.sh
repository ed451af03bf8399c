#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
TAG=${1:-srch}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "search" > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/${TAG}_tests.log; tail -25 gpurun_out/${TAG}_tests.log
: > gpurun_out/${TAG}_bench.txt
for W in url email kwlog; do
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $W --search 2>gpurun_out/${TAG}_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W', d['config']['engine'], d['value'], d['search'])" >> gpurun_out/${TAG}_bench.txt
tail -1 gpurun_out/${TAG}_bench.txt
done
