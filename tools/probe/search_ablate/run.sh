#!/bin/bash
# Where the first-match search kernel's time goes: the kernel rebuilt with phases cut out (copies of the source edited by sed in a
# scratch directory - the product source carries no probe code), timed on one corpus.  usage: run.sh <workload> <bytes>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
W=${1:-url}; N=${2:-8589934592}
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
A='s|^    const uint32_t total_nl = __shfl(incl, 63, 64);|    const uint32_t total_nl = __shfl(incl, 63, 64); { uint32_t x_ = e; for (int i_ = 0; i_ < kEv; i_++) x_ ^= ev[i_]; if (x_ == 0x12345u \&\& total_nl == 0xfffffff0u) match_start[lane] = x_; if (nbytes) continue; }|'
F='s|^    int phase = (vlen == kSearchS \(.*\)$|    int phase = (vlen == kSearchS \1\n    phase = 2;|'
D='s|^        const uint32_t count = fill;$|        const uint32_t count = 0; if (nbytes) { fill = 0; return; }|'
variant "all phases              " ""
variant "forward + numbering only" "$A"
variant "no follow               " "$F"
variant "no walks                " "$D"
variant "no follow, no walks     " "$F;$D"
