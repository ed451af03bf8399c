# the default bench line from a second checkout under .oldtree/ (git archive <commit> | tar -x -C .oldtree; make there) and from this tree, alternating: what a change did to the headline
for t in .oldtree . .oldtree .; do
  (cd $t && timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('$t', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d['config']['table_order_ab_kernel_ms']['ordered'], d['config']['table_order_ab_kernel_ms']['numbered'])")
done
