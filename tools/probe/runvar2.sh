for F in 1 3; do for T in 512 1024; do
  for W in url email; do
    RRX_LIB=$PWD/roaringregex_amd/librrx_f${F}_t$T.so python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $W --bytes 4294967296 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FEED=$F T=$T $W', d['config']['engine'], d['value'], d['roofline']['frac'])"
  done
done; done
