#!/bin/bash
# full verification on the GPU box: the -m gpu suite, smoke(), two fuzz runs (small corpora with the cooperative engines; 4 MiB corpora)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
TAG=${1:-verify}
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/${TAG}_tests.log; tail -8 gpurun_out/${TAG}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log; tail -2 gpurun_out/${TAG}_smoke.log
timeout -k 10 400 python tools/fuzz/gpu_fuzz.py 11 120 > gpurun_out/${TAG}_fuzz_small.txt 2>&1; tail -2 gpurun_out/${TAG}_fuzz_small.txt
timeout -k 10 500 python tools/fuzz/gpu_fuzz.py 12 40 4194304 > gpurun_out/${TAG}_fuzz_big.txt 2>&1; tail -2 gpurun_out/${TAG}_fuzz_big.txt
