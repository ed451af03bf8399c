for a in nondet5000 short5000 short16000; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['config']['engine'], d['config']['stripe_bytes'], d['value'], d['roofline']['kernel_ms_avg'])"
done
