#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
OUT=gpurun_out/$1.txt; : > $OUT
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$LIBTAG $*', d['config']['engine'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['config']['stripe_bytes'])" >> $OUT; tail -1 $OUT; }
for rep in 1 2; do
LIBTAG=base; unset RRX_LIB; run --workload url
LIBTAG=base; run --workload url --stripe 2048
LIBTAG=base; run --workload url --stripe 8192
LIBTAG=base; run --workload url --stripe 16384
LIBTAG=r2; export RRX_LIB=$PWD/roaringregex_amd/librrx_r2.so; run --workload url
unset RRX_LIB
done
