for W in url email arepeat kwlines kwlog; do for E in auto dfa nfa; do
python bench.py --no-cpu-baseline --workload $W --engine $E 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W $E', d['config']['engine'], d['config']['useful_states'], 'GB/s', d['value'], 'frac', d['roofline']['frac'], 'lines', d['config']['lines_per_gpu'], 'acc', d['config']['accepted_rank0'])"
done; done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload url --bytes 2147483648 --pcie 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pcie inclusive GB/s', d['pcie_inclusive_GBs'])"
