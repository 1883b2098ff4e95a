#!/bin/bash
# Round 6: the fused sharded launch (ONE sweep per cycle, the cut-factor workgroups last in its grid waiting for the halo flag) on
# shard 0 of the 8-way cut of configs[3] -- 1 339 cut workgroups, above the default cap -- with CUs kept free for the exchange.
TAG=${1:-r6_shard_fused}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for spec in "0:0:0" "1:1536:0" "1:1536:8" "1:1536:32" "1:1536:64"; do
  IFS=: read f cap cus <<< "$spec"
  MAXSUM_SHARD_FUSED=$f MAXSUM_FUSED_MAX_CUT_BLOCKS=$cap MAXSUM_COMM_CUS=$cus timeout 300 python3 tools/scale_prediction.py --ranks 8 > $OUT/pred_$f_$cap_$cus.log 2>&1
  grep '^{"n"' $OUT/pred_$f_$cap_$cus.log | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fused=$f cap=$cap comm_cus=$cus', d.get('shard_mode'), 'compute', round(d['shard_compute_us'],1), 'loopback cycle', round(d.get('shard_cycle_us_rccl_loopback',-1),1))" || tail -3 $OUT/pred_$f_$cap_$cus.log | cut -c1-300
done 2>&1 | tee $OUT/fused_cus.txt
exit 0
