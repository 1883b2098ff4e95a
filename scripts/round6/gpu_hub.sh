#!/bin/bash
# Round 6: the hub class (kernels.h variable_hub) on the GPU -- parity (hub cases, full-size scale-free instances, the headline
# instance untouched), the AFTER numbers of the scale-free workloads, kernel traces, the A/B against flag 4194304 bounded to 3 cycles.
TAG=${1:-r6_hub}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== parity: hub cases + full-size scale-free + north star"
( time timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "hub or scalefree or north_star or domain300 or wide_hub" --durations=5 ) 2>&1 | tail -14 | tee $OUT/pytest_hub.txt
echo "== AFTER: scale-free colourings"
for wl in coloring_100k_scalefree coloring_1m_scalefree coloring_100k; do
  for dt in f64 f32; do
    timeout 300 python3 bench.py --workload $wl --dtype $dt --steps 200 --warmup 20 --no-cpu-baseline --rows-file $OUT/rows_${wl}_${dt}.json 2>&1 | tail -1 > $OUT/after_${wl}_${dt}.json
    python3 -c "
import json; d=json.loads(open('$OUT/after_${wl}_${dt}.json').read()); print('$wl $dt', round(d['ms_per_step']*1e3,2), 'us/cycle, frac', round(d['roofline']['frac'],4), 'min/max', round(d['timing']['ms_per_step_min']*1e3,2), round(d['timing']['ms_per_step_max']*1e3,2))"
  done
done
echo "== rocprofv3 kernel traces"
cd /tmp
for wl in coloring_100k_scalefree coloring_1m_scalefree; do
  rm -rf $OUT/p
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python3 $R/bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --rows-file /tmp/rows.json > $OUT/prof_$wl.log 2>&1
  f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_${wl}_f64.csv && cut -c1-200 $OUT/kernel_stats_${wl}_f64.csv | head -4
  rm -rf $OUT/p
done
exit 0
