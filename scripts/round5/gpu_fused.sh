#!/bin/bash
# Round 5: the fused sharded launch (one launch per cycle, cut classes last, in-kernel halo wait) re-measured on this round's code.
TAG=${1:-r5_fused}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for f in 0 1; do
  MAXSUM_SHARD_FUSED=$f timeout 600 python tools/scale_prediction.py --ranks 8 > $OUT/pred_fused$f.log 2>&1
  grep '^{"n"' $OUT/pred_fused$f.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fused=$f', d.get('shard_mode'), 'compute', round(d['shard_compute_us'],1), 'loopback', round(d.get('shard_cycle_us_rccl_loopback',-1),1), d.get('predicted_speedup_vs_one_gpu'))"
done 2>&1 | tee $OUT/fused_ab.txt
MAXSUM_COST_ONLY=bd timeout 300 python tools/shard_cost.py 8 f64 2>&1 | grep "^{" | tail -1 | cut -c1-900 | tee $OUT/shard_cost_weak8.json
exit 0
