#!/bin/bash
# SURVEY 8(d) long-line variants of C2/C3: parity test, then the bench line with --search for both.  usage: r04_long_lines.sh <tag>
set -o pipefail
tag=${1:-r04_long}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -x -k "long_line_variants" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for w in email_long url_long; do
  timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 --search > $out/$w.json 2> $out/$w.err || { tail -20 $out/$w.err; exit 1; }
  python3 - $out/$w.json <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
c = d["config"]
print(c["workload"][:60], "| match", d["value"], d["unit"], "frac", d["roofline"]["frac"], "| one-shot", d.get("cold", {}).get("one_shot_frac_of_peak"), "| search", {k: v for k, v in c.items() if "search" in k})
PY
done
