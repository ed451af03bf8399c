#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
O=gpurun_out/units_ab.txt; : > $O
G=$((1<<30)); M1=$((1<<20)); M2=$((2<<20))
timeout -k 10 300 python tools/probe/units/ab.py arepeat $G 0,0 4096,$((M2+16)) 4096,16 2048,32 2048,64 1024,64 >> $O 2>&1 || { tail -5 $O; exit 1; }
timeout -k 10 300 python tools/probe/units/ab.py email $G 0,0 2048,$((M2+16)) 2048,16 1024,32 1024,64 512,64 512,128 >> $O 2>&1 || { tail -5 $O; exit 1; }
timeout -k 10 300 python tools/probe/units/ab.py url $G 0,0 2048,16 1024,32 1024,64 512,64 512,128 >> $O 2>&1 || { tail -5 $O; exit 1; }
timeout -k 10 400 python tools/probe/units/ab.py url $((8*G)) 0,0 4096,$((M2+16)) 4096,16 4096,32 2048,32 2048,64 1024,64 1024,128 >> $O 2>&1 || { tail -5 $O; exit 1; }
grep -v amdgpu.ids $O
