#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
OUT=gpurun_out/$1.txt; : > $OUT
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['config']['engine'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['config']['stripe_bytes'])" >> $OUT; tail -1 $OUT; }
for S in 0 1024 2048; do for E in auto dfa; do run --workload kwlines --stripe $S --engine $E; done; done
for S in 0 1024 2048; do run --workload email --stripe $S; done
