#!/bin/bash
# rocprofv3 kernel traces of the configurations whose layout changed with the tiled factor order (closing code),
# and of the amaxsum run
TAG=${1:-r4_traces2}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for spec in coloring_100k:f32:2000 coloring_1m_deg6:f32:300 coloring_1m_deg6:f64:300; do
  IFS=: read w dt st <<< "$spec"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/bench.py --no-cpu-baseline --configs main --workload $w --dtype $dt --steps $st --warmup $((st/10)) > $OUT/prof_${w}_${dt}.log 2>&1
  f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_${w}_${dt}.csv; echo "$w $dt: $(sed -n 2p $f | cut -c1-150)"; fi
  rm -rf $OUT/p
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/tools/amaxsum_bench.py --no-oracle 100000 > $OUT/prof_amaxsum.log 2>&1
f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_amaxsum100k.csv && cut -c1-140 $OUT/kernel_stats_amaxsum100k.csv | head -8
rm -rf $OUT/p
