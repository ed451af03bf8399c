for i in 1 2; do for S in 2048 4096 8192 16384; do
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload url --stripe $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('url 8GiB stripe $S', d['value'])"
done; done
for S in 2048 4096 8192 16384; do
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload kwlog --stripe $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kwlog 8GiB stripe $S', d['value'])"
done
