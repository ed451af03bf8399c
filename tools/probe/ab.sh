# usage: ab.sh libA libB [bench args]  — interleaved A/B on one box
A=$1; B=$2; shift 2
for i in 1 2; do for L in $A $B; do
RRX_LIB=$PWD/roaringregex_amd/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['config']['engine'], d['value'])"
done; done
