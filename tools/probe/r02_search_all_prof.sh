#!/bin/bash
# Kernel trace of bench.py --search: the three search modes and the one-launch all-matches kernel side by side.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-url}
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_search_all -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --search --workload $W > $R/gpurun_out/prof_search_all.log 2>&1
head -30 $R/gpurun_out/prof_search_all/t_kernel_stats.csv | cut -c1-200
