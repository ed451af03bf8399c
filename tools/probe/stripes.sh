for S in 1024 2048 4096 8192 16384; do
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload email --stripe $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('email 1GiB stripe $S', d['value'])"
done
for S in 4096 8192 16384; do
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload url --bytes 2147483648 --stripe $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('url 2GiB stripe $S', d['value'])"
done
