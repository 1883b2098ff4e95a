#!/bin/bash
# Sharded-cycle measurements on one GPU: parity of k shards, cost of one rank's cycle with the
# fused launch (default) and with the two-launch schedule, kernel trace of the fused run.
# usage: gpurun --timeout 400 -- 'bash scripts/gpu_shard.sh TAG'
TAG=${1:-shard}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
T0=$(date +%s); lap() { echo "== [$(( $(date +%s) - T0 ))s] $1"; }
lap "pytest -m gpu (sharded)"
timeout 300 python -m pytest tests/test_sharded.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest_sharded.txt
lap "shard cost: direct exchange (default for the native path)"
(timeout 120 python tools/shard_cost.py 8 f64 2>&1 | tail -1; timeout 120 python tools/shard_cost.py 2 f64 2>&1 | tail -1) | tee $OUT/shard_cost_direct.jsonl
lap "shard cost: every factor class in the second launch (layout flag 512)"
(MAXSUM_LAYOUT_FLAGS=512 timeout 120 python tools/shard_cost.py 8 f64 2>&1 | tail -1; MAXSUM_LAYOUT_FLAGS=512 timeout 120 python tools/shard_cost.py 2 f64 2>&1 | tail -1) | tee $OUT/shard_cost_factors_second.jsonl
export MAXSUM_LAYOUT_FLAGS=512
lap "kernel trace of the shard cycle (8-way shard 0, direct exchange, flag 512)"
( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $R/tools/shard_cost.py 8 f64 > $OUT/prof.log 2>&1 )
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | while read f; do head -12 "$f" | cut -c1-200; cp "$f" $OUT/kernel_stats_shard.csv; done
rm -rf $OUT/prof
unset MAXSUM_LAYOUT_FLAGS
lap "bench default"
timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 | tee $OUT/bench_default.json
lap done
