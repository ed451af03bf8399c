#!/bin/bash
# round-4 record: headline profile (kernel trace + PMC), search profiles, sampled-table profile, all configs x engines
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
bash tools/probe/prof.sh r04_headline > gpurun_out/r04_headline_summary.txt 2>&1 || { tail -5 gpurun_out/r04_headline_summary.txt; exit 1; }
bash tools/probe/r04_search_prof.sh r04_search_url url 8589934592 > gpurun_out/r04_search_url.log 2>&1 || { tail -5 gpurun_out/r04_search_url.log; exit 1; }
bash tools/probe/r04_search_prof.sh r04_search_email email 1073741824 > gpurun_out/r04_search_email.log 2>&1 || { tail -5 gpurun_out/r04_search_email.log; exit 1; }
bash tools/probe/sampled_prof.sh > gpurun_out/r04_sampled.log 2>&1 || { tail -5 gpurun_out/r04_sampled.log; exit 1; }
bash tools/probe/r02_run.sh r04_all notests "auto dfa nfa" > gpurun_out/r04_all.log 2>&1 || { tail -5 gpurun_out/r04_all.log; exit 1; }
for a in "--workload nondet" "--workload nondet600" "--workload nondet5000" "--workload urltail" "--workload urlalt"; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $a 2>>gpurun_out/r04_all_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['config']['engine'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['config']['accepted_rank0'])" >> gpurun_out/r04_all_bench.txt
done
bash tools/probe/r04_search.sh r04_final notests > gpurun_out/r04_search_final.log 2>&1
tail -30 gpurun_out/r04_all_bench.txt; cat gpurun_out/r04_search_final.log
