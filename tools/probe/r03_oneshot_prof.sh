#!/bin/bash
# kernel trace of one bench.py run -> per-call breakdown of the twelve one-shot calls.  usage: r03_oneshot_prof.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/kt.err || { tail -5 $OUT/kt.err; exit 1; }
python3 tools/probe/one_shot_calls.py $OUT/kt $OUT/bench.json > $OUT/one_shot_calls.txt
cat $OUT/one_shot_calls.txt
