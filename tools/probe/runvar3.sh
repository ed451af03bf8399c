for S in 4096 8192 16384; do for F in 1 3; do for W in url email; do
RRX_LIB=$PWD/roaringregex_amd/librrx_s${S}_f$F.so python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $W --bytes 4294967296 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('STRIPE=$S FEED=$F $W', d['value'], d['roofline']['frac'])"
done; done; done
