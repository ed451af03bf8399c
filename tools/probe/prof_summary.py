"""Summarise the rocprofv3 CSVs written by prof.sh into one small text file (committed under profiles/)."""
import csv, glob, os, sys
out = sys.argv[1]
lines = []
def find(d, pat):
    return glob.glob(os.path.join(out, d, '**', pat), recursive=True)
for f in find('kt', '*kernel_stats.csv'):
    lines.append('== kernel stats (%s)' % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        lines.append('  %-90s calls %s  total_ns %s  avg_ns %s  pct %s' % (row.get('Name', '')[:90], row.get('Calls'), row.get('TotalDurationNs'), row.get('AverageNs'), row.get('Percentage')))
for d in ('pmc_sq', 'pmc_sq2', 'pmc_fetch', 'pmc_write'):
    for f in find(d, '*counter_collection.csv'):
        agg = {}
        for row in csv.DictReader(open(f)):
            k = (row.get('Kernel_Name', '')[:70], row.get('Counter_Name'))
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1; a[1] += float(row.get('Counter_Value', 0))
        lines.append('== counters (%s): mean per dispatch' % d)
        for (kn, cn), (n, s) in sorted(agg.items()):
            if "match_" in kn or "count_newlines" in kn or "search_" in kn:
                lines.append('  %-70s %-24s n=%d mean=%.6g' % (kn, cn, n, s / n))
txt = '\n'.join(lines)
open(os.path.join(out, 'summary.txt'), 'w').write(txt + '\n')
print(txt)
