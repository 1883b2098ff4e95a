#!/bin/bash
# amaxsum: GPU parity tests, messages per second, kernel trace of the 100k-variable run
TAG=${1:-amaxsum}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
( time timeout 500 python -m pytest tests/test_gpu_amaxsum.py -x -q -m gpu --durations=4 ) 2>&1 | tail -12 | tee $OUT/pytest.txt
timeout 200 python tools/amaxsum_bench.py --no-oracle 10000 100000 | tee $OUT/amaxsum_bench.jsonl
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/tools/amaxsum_bench.py --no-oracle 100000 > $OUT/prof.log 2>&1
f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_amaxsum100k.csv && cut -c1-160 $OUT/kernel_stats_amaxsum100k.csv | head -12; rm -rf $OUT/p
