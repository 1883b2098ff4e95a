#!/bin/bash
# Shard cycle (8-way shard 0, RCCL loopback) with CUs reserved for the comm stream:
# two-launch and fused schedules.  usage: gpurun --timeout 300 -- 'bash scripts/gpu_shard2.sh TAG'
TAG=${1:-shard2}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
export MAXSUM_COST_ONLY=b
: > $OUT/shard_cost_cus.jsonl
for fused in 0 1; do for cus in 0 8 16 32; do
  MAXSUM_SHARD_FUSED=$fused MAXSUM_COMM_CUS=$cus timeout 100 python tools/shard_cost.py 8 f64 2>&1 | grep "^{" | tail -1 | tee -a $OUT/shard_cost_cus.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fused', '$fused', 'comm cus', '$cus', d.get('shard_mode'), 'us/cycle', round(d.get('shard_cycle_us_native_rccl_loopback', -1), 2))"
done; done
