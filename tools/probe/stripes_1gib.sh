for W in email arepeat; do for S in 1024 2048 4096; do for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $W --stripe $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W stripe $S', d['value'])"
done; done; done
