for T in 256 512 1024; do
  for W in url email; do
    RRX_LIB=$PWD/roaringregex_amd/librrx_t$T.so python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $W --bytes 4294967296 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T=$T', d['config']['engine'], '$W', d['value'], d['roofline']['frac'])"
  done
done
RRX_LIB=$PWD/roaringregex_amd/librrx_t1024.so python bench.py --steps 5 --warmup 1 --no-cpu-baseline --engine nfa --bytes 4294967296 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T=1024 nfa url', d['value'])"
