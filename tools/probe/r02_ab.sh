#!/bin/bash
# usage: r02_ab.sh <tag> "<lib suffixes>" "<workloads>" "<engines>" — A/B of library builds (same ABI) through bench.py
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
TAG=$1; LIBS=$2; WORKLOADS=$3; ENGINES=${4:-nfa}; EXTRA=$5
mkdir -p gpurun_out
: > gpurun_out/${TAG}_ab.txt
for W in $WORKLOADS; do for E in $ENGINES; do for L in $LIBS; do
LIBF=$PWD/roaringregex_amd/librrx$L.so
RRX_LIB=$LIBF timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload $W --engine $E $EXTRA 2>gpurun_out/${TAG}_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W $E lib[$L]', d['config']['engine'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['config']['accepted_rank0'])" >> gpurun_out/${TAG}_ab.txt
tail -1 gpurun_out/${TAG}_ab.txt
done; done; done
