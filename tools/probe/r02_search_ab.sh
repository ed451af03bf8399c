#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
: > gpurun_out/$1.txt
for L in "" _snf _sne _snfe ""; do
RRX_LIB=$PWD/roaringregex_amd/librrx$L.so python3 tools/probe/search_run.py url 4294967296 5 2>/dev/null | sed "s/^/lib[$L] /" >> gpurun_out/$1.txt
tail -1 gpurun_out/$1.txt
done
