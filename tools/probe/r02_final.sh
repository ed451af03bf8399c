#!/bin/bash
# round-2 measurement set: default bench line (with cpu_baseline), every workload x engine, the headline's rocprofv3 passes
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
TAG=${1:-r02_final}
python bench.py > gpurun_out/${TAG}_default.json 2> gpurun_out/${TAG}_default.err
tail -c 600 gpurun_out/${TAG}_default.json
bash tools/probe/r02_run.sh $TAG notests "auto dfa nfa" "url email arepeat kwlines kwlog nondet"
for A in "--workload nondet600 --engine auto" "--workload url --engine wave --bytes 1073741824" "--workload nondet --engine wave --bytes 1073741824"; do
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$A', d['config']['engine'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['config']['accepted_rank0'])" >> gpurun_out/${TAG}_bench.txt
tail -1 gpurun_out/${TAG}_bench.txt
done
bash tools/probe/prof.sh ${TAG}_url --workload url > /dev/null 2>&1
cat gpurun_out/prof_${TAG}_url/summary.txt | grep -E "match_stripes2|count_newlines|== " | head -30
