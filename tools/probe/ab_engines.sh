for W in "$@"; do for i in 1 2; do for E in dfa dfa2; do
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload $W --engine $E 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W $E', d['config']['engine'], d['value'], d['roofline']['frac'], d['config']['accepted_rank0'])"
done; done; done
