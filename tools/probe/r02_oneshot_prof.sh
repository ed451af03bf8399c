#!/bin/bash
# Kernel trace of the one-shot entry (rrx_match_device): where the time beyond the match kernel goes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for W in email url; do
  rocprofv3 --kernel-trace -d $R/gpurun_out/prof_oneshot_$W -o t -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --workload $W > $R/gpurun_out/prof_oneshot_$W.log 2>&1
done
