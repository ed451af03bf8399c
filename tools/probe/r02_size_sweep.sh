#!/bin/bash
# Kernel time against corpus size (email config): where the fixed cost of a launch goes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 268435456 536870912 1073741824 2147483648 4294967296; do
  for S in 2048 4096; do
    timeout -k 10 200 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload email --bytes $B --stripe $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($B, $S, d[\"value\"], d[\"roofline\"][\"frac\"], d[\"roofline\"][\"kernel_ms_avg\"])"
  done
done
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_email1g -o email1g -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload email > $R/gpurun_out/prof_email1g.log 2>&1
find $R/gpurun_out/prof_email1g -name "*kernel_stats.csv" | xargs head -5
