#!/bin/bash
# One PMC pass (counters in $1, space separated) over tools/microbench and selected bench workloads.
# usage: gpu_pmc.sh TAG "COUNTER1 COUNTER2" workload1 workload2 ...
TAG=$1; shift; CTRS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/cal -o pmc -- $R/tools/microbench > $OUT/cal.log 2>&1
f=$(find $OUT/cal -name "*counter_collection*.csv" | head -1)
[ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" | grep -v -E "k_empty|k_chase|rocclr" | tee $OUT/calibration.txt
rm -rf $OUT/cal
for w in "$@"; do
  timeout 600 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p_$w -o pmc -- python $R/bench.py --no-cpu-baseline --workload $w --steps 30 --warmup 10 --graph-chunk 0 > $OUT/$w.log 2>&1
  f=$(find $OUT/p_$w -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" | grep -v rocclr | sed "s/^/$w /" | tee $OUT/$w.txt
  rm -rf $OUT/p_$w
done
