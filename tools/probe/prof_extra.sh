#!/bin/bash
# rocprofv3 kernel statistics of the two paths beside the batch matcher: search (bench.py --search) and one long string
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/prof_extra
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/search -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --search > $OUT/search.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/longstring -- python3 tools/probe/longstring.py > $OUT/longstring.log 2>&1
