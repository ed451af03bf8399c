# flush cost: p0 = product, p1 = no flush at all, p4 = plain store instead of the atomic, p5 = flush arithmetic only
for W in "$@"; do for L in librrx_p0.so librrx_p6.so librrx_p5.so librrx_p0.so; do
RRX_LIB=$PWD/roaringregex_amd/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W $L', d['config']['engine'], d['value'])"
done; done
