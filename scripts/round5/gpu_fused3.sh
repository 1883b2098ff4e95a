#!/bin/bash
# Round 5: is the fused sharded launch (opt-in) faster than two launches where it qualifies?  Shard 0 of the 8-way cut of the
# weak-scaling instance (8 x 100k variables), RCCL loopback, fused on / off; and the peer-store loopback.
TAG=${1:-r5_fused3}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for f in 0 1; do
  MAXSUM_SHARD_FUSED=$f MAXSUM_COST_ONLY=b timeout 300 python tools/shard_cost.py 8 f64 2>&1 | grep "^{" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fused=$f', d.get('shard_mode'), 'rccl loopback', round(d.get('shard_cycle_us_native_rccl_loopback',-1),1), 'factors', d['shard_factors'])"
done 2>&1 | tee $OUT/fused_weak8.txt
exit 0
