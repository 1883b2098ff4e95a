#!/bin/bash
# Peer-store exchange: publish kernel on the comm stream (default) vs inline on the compute stream.
TAG=${1:-shard4}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_sharded.py -x -q -m gpu -k "peer or nccl" 2>&1 | tail -3 | tee $OUT/pytest_sharded.txt
export MAXSUM_COST_ONLY=d
: > $OUT/shard_cost_publish.jsonl
for mode in comm inline; do for n in 8 2; do
  MAXSUM_P2P_PUBLISH=$mode timeout 100 python tools/shard_cost.py $n f64 2>&1 | grep "^{" | tail -1 | tee -a $OUT/shard_cost_publish.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('publish', '$mode', 'ranks', d['ranks'], 'us/cycle', round(d.get('shard_cycle_us_peer_stores_loopback', -1), 2))"
done; done
