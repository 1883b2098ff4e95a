#!/bin/bash
# Peer-store exchange on one GPU: parity (two processes sharing the GPU, real hipIpc) and the
# cost of one rank's cycle with the exchange looped back, next to the RCCL loopback.
TAG=${1:-shard3}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
T0=$(date +%s); lap() { echo "== [$(( $(date +%s) - T0 ))s] $1"; }
lap "pytest -m gpu (sharded)"
timeout 400 python -m pytest tests/test_sharded.py -x -q -m gpu --durations=4 2>&1 | tail -12 | tee $OUT/pytest_sharded.txt
lap "shard cost: RCCL loopback (direct) and peer stores looped back"
export MAXSUM_COST_ONLY=bd
(timeout 120 python tools/shard_cost.py 8 f64 2>&1 | grep "^{" | tail -1; timeout 120 python tools/shard_cost.py 2 f64 2>&1 | grep "^{" | tail -1; timeout 120 python tools/shard_cost.py 4 f64 2>&1 | grep "^{" | tail -1) | tee $OUT/shard_cost_p2p.jsonl
lap "kernel trace, peer stores looped back (8-way shard 0)"
export MAXSUM_COST_ONLY=d
( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $R/tools/shard_cost.py 8 f64 > $OUT/prof.log 2>&1 )
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | while read f; do head -8 "$f" | cut -c1-200; cp "$f" $OUT/kernel_stats_p2p.csv; done
rm -rf $OUT/prof
unset MAXSUM_COST_ONLY
lap "bench default"
timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 | tee $OUT/bench_default.json
lap done
