#!/bin/bash
# rocprofv3 kernel-trace stats of one bench variant.  usage: gpu_prof.sh TAG "bench args"
TAG=${1:-prof}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/bench.py --no-cpu-baseline $@ > $OUT/prof.log 2>&1
tail -1 $OUT/prof.log | cut -c1-200
f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); cp "$f" $OUT/kernel_stats.csv; rm -rf $OUT/p
cut -c1-220 $OUT/kernel_stats.csv | head -12
