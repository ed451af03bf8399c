#!/bin/bash
# round-2 baseline on this round's box: gpu tests, then every workload x engine through bench.py
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_base_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r02_base_tests.log
tail -3 gpurun_out/r02_base_tests.log
: > gpurun_out/r02_base_bench.txt
for W in url email arepeat kwlines kwlog; do for E in auto dfa nfa; do
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $W --engine $E 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W $E', d['config']['engine'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['config']['accepted_rank0'], d['setup'])" >> gpurun_out/r02_base_bench.txt
tail -1 gpurun_out/r02_base_bench.txt
done; done
