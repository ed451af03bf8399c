cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/fetch_kw; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --workload kwlines > $OUT/f.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/f/**/*counter_collection.csv", recursive=True):
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "match_stripes2_kernel" in r["Kernel_Name"] or "match_stripes2_short" in r["Kernel_Name"]]
    print("batch kernel FETCH_SIZE x2 = %.3f GB over %d launches (text 8.590 GB)" % (sum(v)/len(v)*1024*2/1e9, len(v)))
PY
