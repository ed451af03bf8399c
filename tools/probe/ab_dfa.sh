# usage: ab_dfa.sh libA libB engine workloads...  — interleaved A/B of one engine on one box
A=$1; B=$2; E=$3; shift 3
for W in "$@"; do for i in 1 2; do for L in $A $B; do
RRX_LIB=$PWD/roaringregex_amd/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline --workload $W --engine $E 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W $L', d['config']['engine'], d['value'])"
done; done; done
