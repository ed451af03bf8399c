#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/nfa3_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/nfa3_tests.log; tail -12 gpurun_out/nfa3_tests.log
bash tools/probe/r02_run.sh nfa3 notests "nfa" "url email arepeat kwlines kwlog nondet"
bash tools/probe/prof.sh r02_nfa_url --workload url --engine nfa > /dev/null 2>&1
bash tools/probe/prof.sh r02_nfa_arepeat --workload arepeat --engine nfa > /dev/null 2>&1
bash tools/probe/prof.sh r02_nfa_nondet --workload nondet --engine nfa > /dev/null 2>&1
ls gpurun_out/prof_r02_nfa_url/ | head
