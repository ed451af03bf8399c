for T in 256 512 1024; do for W in url email; do
RRX_LIB=$PWD/roaringregex_amd/librrx_t$T.so python bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T=$T $W', d['value'], d['roofline']['frac'])"
done; done
