#!/bin/bash
# the follow loop with four bytes per turn (round 3) against sixteen (round 4), per mode.  usage: follow_ab.sh <workload> <bytes> ...
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
S=/tmp/ablate; rm -rf $S; mkdir -p $S; cp -r $R/roaringregex_amd $R/include $R/tools $R/bench.py $R/tests $S/ 2>/dev/null
cd $S/roaringregex_amd/csrc
SRC=kernels_search.hip
cp $SRC $SRC.orig
build() {
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wno-unused-parameter --offload-arch=gfx950 -c $SRC -o build/$SRC.o 2> /tmp/ablate_cc.log || { tail -5 /tmp/ablate_cc.log; exit 1; }
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o ../librrx.so build/*.o || exit 1
}
run() { while [ $# -ge 2 ]; do (cd $S && timeout -k 10 200 python3 tools/probe/search_ablate/modes.py $1 $2) || exit 1; shift 2; done; }
echo "== sixteen bytes per turn"; build; run "$@"
python3 - <<PY
s = open("$SRC.orig").read()
a = s.index("    size_t fbyte = my_end;"); b = s.index("        // ---- 4c. the walks still waiting")
open("$SRC", "w").write(s[:a] + open("$S/tools/probe/search_ablate/follow_4_bytes_per_turn.inc").read() + s[b:])
PY
echo "== four bytes per turn"; build; run "$@"
