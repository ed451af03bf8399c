#!/bin/bash
# usage: r02_run.sh <tag> [tests|notests] [engine list] — gpu tests (optional), then workloads x engines through bench.py
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
TAG=$1; DO_TESTS=${2:-tests}; ENGINES=${3:-"auto nfa"}; WORKLOADS=${4:-"url email arepeat kwlines kwlog"}
mkdir -p gpurun_out
if [ "$DO_TESTS" = tests ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1
  echo "tests rc=$?" >> gpurun_out/${TAG}_tests.log
  tail -15 gpurun_out/${TAG}_tests.log
fi
: > gpurun_out/${TAG}_bench.txt
for W in $WORKLOADS; do for E in $ENGINES; do
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $W --engine $E 2>gpurun_out/${TAG}_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W $E', d['config']['engine'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['config']['accepted_rank0'], d['cold'])" >> gpurun_out/${TAG}_bench.txt
tail -1 gpurun_out/${TAG}_bench.txt
done; done
