for cfg in "email_long 2048" "email_long 4096" "email_long 8192" "url_long 4096" "url_long 8192" "url_long 16384" "arepeat 2048" "arepeat 4096"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --workload $1 --stripe $2 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('$1', '$2', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'])"
done
