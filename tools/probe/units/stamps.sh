#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
O=gpurun_out/units_stamps.txt; : > $O
G=$((1<<30)); M2=$((2<<20))
for v in 0 16 $((M2+16)); do
  timeout -k 10 200 python tools/probe/stamps/run.py arepeat $G 4096 $v >> $O 2>&1 || { tail -5 $O; exit 1; }
done
grep -v "ten \|amdgpu.ids" $O
