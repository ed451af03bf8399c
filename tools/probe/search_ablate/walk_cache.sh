#!/bin/bash
# How much of the walks' time is text that has left the caches?  The first-match kernel rebuilt with the walks reading their text from the
# FIRST 4 KiB of the corpus (always in the L1/L2: the results are wrong, the turn counts those of the same kind of text), against the
# shipping kernel and against "no walks".  usage: walk_cache.sh <workload> <bytes>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
W=${1:-email}; N=${2:-1073741824}
S=/tmp/ablate; rm -rf $S; mkdir -p $S; cp -r $R/roaringregex_amd $R/include $R/tools $R/bench.py $R/tests $S/ 2>/dev/null
cd $S/roaringregex_amd/csrc
SRC=kernels_search.hip
cp $SRC $SRC.orig
variant() {  # name, sed script
  cp $SRC.orig $SRC
  [ -n "$2" ] && sed -i "$2" $SRC
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wno-unused-parameter --offload-arch=gfx950 -c $SRC -o build/$SRC.o 2> /tmp/ablate_cc.log || { tail -5 /tmp/ablate_cc.log; exit 1; }
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o ../librrx.so build/*.o || exit 1
  echo -n "$1: "; (cd $S && timeout -k 10 120 python3 tools/probe/search_run.py $W $N 5) || exit 1
}
C='s|^    auto text_word = \[&\](uint32_t at) -> uint32_t { return \*reinterpret_cast<const uint32_t \*>(cbase + at); };.*$|    auto text_word = [\&](uint32_t at) -> uint32_t { return *reinterpret_cast<const uint32_t *>(bytes + (at \& 0xffcu)); };|'
D='s|^        const uint32_t count = fill;$|        const uint32_t count = 0; if (nbytes) { fill = 0; return; }|'
variant "all phases                      " ""
variant "walks read cached text (wrong!) " "$C"
variant "no walks                        " "$D"
