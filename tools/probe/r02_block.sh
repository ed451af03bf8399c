#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/blk_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/blk_tests.log; tail -12 gpurun_out/blk_tests.log
bash tools/probe/prof.sh r02_group_nondet600 --workload nondet600 > /dev/null 2>&1
cat gpurun_out/prof_r02_group_nondet600/summary.txt | head -30
