#!/bin/bash
# PMC passes, ONE counter per pass (FETCH_SIZE + WRITE_SIZE together exceed what the hardware
# collects in one pass: rocprofv3 aborts), short timeouts.
# usage: gpu_pmc.sh TAG "COUNTER ..." "workload:dtype:flags ..."
TAG=$1; CTRS=$2; SPECS=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for spec in $SPECS; do
  IFS=: read w dt fl <<< "$spec"
  for c in $CTRS; do
    t0=$(date +%s)
    timeout 200 rocprofv3 --pmc $c --output-format csv -d $OUT/p -o pmc -- python $R/bench.py --no-cpu-baseline --configs main --workload $w --dtype $dt --steps 20 --warmup 5 --graph-chunk 0 --layout-flags $fl > $OUT/log_${w}_${dt}_${fl}_$c.txt 2>&1
    f=$(find $OUT/p -name "*counter_collection*.csv" | head -1)
    if [ -n "$f" ]; then python $R/scripts/pmc_summary.py "$f" | grep -v rocclr | sed "s/^/$w $dt flags=$fl /" | tee -a $OUT/pmc_${w}_${dt}_${fl}.txt; rm -f $OUT/log_${w}_${dt}_${fl}_$c.txt; else echo "FAILED $spec $c"; tail -3 $OUT/log_${w}_${dt}_${fl}_$c.txt; fi
    rm -rf $OUT/p
    echo "  ($(( $(date +%s) - t0 )) s)"
  done
done
