for cfg in "url 1024" "url 2048" "url 4096" "kwlines 1024" "kwlines 2048" "kwlog 1024" "kwlog 4096"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --workload $1 --stripe $2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('$1', '$2', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'])"
done
