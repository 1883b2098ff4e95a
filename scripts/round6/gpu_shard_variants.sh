#!/bin/bash
# Round 6: shard 0 of the 8-way cut of configs[3] -- which schedule of a sharded cycle is the fastest one GPU can show?
#   flags 0      launch 1 = variables + interior factors, launch 2 = cut factors (the default)
#   flags 512    launch 1 = variables only, launch 2 = every factor class (the exchange may start after the variables)
#   MAXSUM_COMM_CUS=n   the compute stream leaves n CUs to the comm stream's kernels (RCCL)
TAG=${1:-r6_shard_variants}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for cus in 0 8 16; do for fl in 0 512; do
  MAXSUM_COMM_CUS=$cus timeout 400 python3 tools/scale_prediction.py --ranks 8 --layout-flags $fl > $OUT/pred_cus${cus}_fl$fl.log 2>&1
  grep '^{"n"' $OUT/pred_cus${cus}_fl$fl.log | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('comm_cus=$cus flags=$fl', d.get('shard_mode'), 'compute', round(d['shard_compute_us'],1), 'loopback cycle', round(d.get('shard_cycle_us_rccl_loopback',-1),1), d.get('predicted_speedup_vs_one_gpu'))"
done; done 2>&1 | tee $OUT/variants.txt
exit 0
