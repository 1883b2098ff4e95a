#!/bin/bash
# Round 5: the fused sharded launch on shard 0 of the 8-way cut of configs[3] with the cap on waiting cut blocks raised (1 339 cut blocks).
# (Provenance only: the measurement build honoured $MAXSUM_FUSED_MAX_CUT_BLOCKS; the parked workgroups starved the exchange -- the
# 5-s in-kernel timeout fired -- and the override was removed again)
TAG=${1:-r5_fused2}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for f in 1 0; do
  MAXSUM_SHARD_FUSED=$f MAXSUM_FUSED_MAX_CUT_BLOCKS=1536 timeout 300 python tools/scale_prediction.py --ranks 8 > $OUT/pred_fused$f.log 2>&1
  grep '^{"n"' $OUT/pred_fused$f.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fused=$f', d.get('shard_mode'), 'compute', round(d['shard_compute_us'],1), 'loopback', round(d.get('shard_cycle_us_rccl_loopback',-1),1), d.get('predicted_speedup_vs_one_gpu'))"
  tail -3 $OUT/pred_fused$f.log | cut -c1-300
done 2>&1 | tee $OUT/fused_ab.txt
exit 0
