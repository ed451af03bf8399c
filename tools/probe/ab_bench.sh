#!/bin/bash
# A/B on ONE box: tools/probe/ab/librrx_base.so (a copy of an earlier build) against the in-tree librrx.so, alternating.
# usage: ab_bench.sh <tag> <rounds> <bench args...>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
TAG=$1; N=$2; shift; shift
for i in $(seq 1 $N); do
  for V in base new; do
    if [ $V = base ]; then export RRX_LIB=$R/tools/probe/ab/librrx_base.so; else unset RRX_LIB; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', '$*', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], 'oneshot', d['cold']['one_shot_ms'])" >> gpurun_out/${TAG}.txt || exit 1
    tail -1 gpurun_out/${TAG}.txt
  done
done
