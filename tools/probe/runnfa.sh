for W in url email arepeat kwlines kwlog; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $W --engine nfa 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W nfa', 'GB/s', d['value'], 'acc', d['config']['accepted_rank0'])"
done
