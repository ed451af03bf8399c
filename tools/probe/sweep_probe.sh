for L in p1 p2 p3 p4; do
RRX_LIB=$PWD/roaringregex_amd/librrx_$L.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['roofline']['frac'])"
done
