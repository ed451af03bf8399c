#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/grp_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/grp_tests.log; tail -12 gpurun_out/grp_tests.log
: > gpurun_out/grp_bench.txt
run() { timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>>gpurun_out/grp_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['config']['engine'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['config']['accepted_rank0'])" >> gpurun_out/grp_bench.txt; tail -1 gpurun_out/grp_bench.txt; }
run --workload nondet --engine auto
run --workload nondet600 --engine auto
run --workload url --engine wave --bytes 1073741824
run --workload kwlog --engine wave --bytes 1073741824
