for L in librrx.so librrx_occ8.so; do for W in url email; do
RRX_LIB=$PWD/roaringregex_amd/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L $W', d['value'], d['roofline']['frac'])"
done; done
