#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "one_shot or kat_batch or golden" > gpurun_out/op_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/op_tests.log; tail -25 gpurun_out/op_tests.log
bash tools/probe/r02_run.sh op notests "auto" "url email arepeat kwlines kwlog"
