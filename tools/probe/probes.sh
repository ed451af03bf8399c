# p0 = product, p1 = no P gather, p2 = no T2 gather, p3 = neither (results of p1..p3 are wrong)
for W in "$@"; do for L in librrx_p0.so librrx_p1.so librrx_p2.so librrx_p3.so librrx_p0.so; do
RRX_LIB=$PWD/roaringregex_amd/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W $L', d['config']['engine'], d['value'])"
done; done
