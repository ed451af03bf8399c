#!/bin/bash
# rocprofv3 passes for the batch kernel (kernel trace; SQ/LDS counters; HBM bytes).  Usage: prof.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline "$@" > $OUT/kt.log 2>&1 || exit 1
timeout -k 10 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline "$@" > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline "$@" > $OUT/pmc_sq2.log 2>&1 || exit 1
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1 || exit 1
python3 tools/probe/prof_summary.py $OUT
