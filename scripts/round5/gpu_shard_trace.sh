#!/bin/bash
# Round 5: where a shard's cycle goes -- shard 0 of the 8-way cut of BASELINE configs[3] on one GPU: kernel trace of
# both launches of a sharded cycle (tools/scale_prediction.py --ranks 8: (a) compute + pack / unpack, (b) RCCL loopback),
# then FETCH_SIZE / WRITE_SIZE of the same.
TAG=${1:-r5_shard_trace}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for fl in ${@:-0}; do
rm -rf /tmp/prof_shard
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_shard -o t -- python $R/tools/scale_prediction.py --ranks 8 --layout-flags $fl > $OUT/pred_$fl.log 2>&1
f=$(find /tmp/prof_shard -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_shard0_of_8_flags$fl.csv; echo "-- flags $fl"; head -12 "$f" | cut -d, -f1-4,6,7 | cut -c1-200; fi
grep '^{"n"' $OUT/pred_$fl.log | tail -1 | cut -c1-600
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_shard
  timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_shard -o pmc -- python $R/tools/scale_prediction.py --ranks 8 > /dev/null 2>&1
  f=$(find /tmp/pmc_shard -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" | grep -v rocclr | tee -a $OUT/pmc_shard0_of_8.txt
done
exit 0
