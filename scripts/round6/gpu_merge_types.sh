#!/bin/bash
# Round 6: lane-grid groups of one shape sharing the wider sibling's storage type (layout.cpp, BIN2_MERGE_BYTES; layout_flags
# bit24 = 16777216 keeps them split): cycle times with / without, parity of the rows it changes, serial kernel traces.
TAG=${1:-r6_merge}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for spec in secp_100k:f64 secp_100k:f32 secp_100k_m4:f64 secp_100k_m4:f32 peav_50k:f64 peav_50k:f32 coloring_100k_d8:f64 meeting_50k_hetero:f64; do
  IFS=: read wl dt <<< "$spec"
  for fl in 16777216 0; do
    timeout 300 python3 bench.py --workload $wl --dtype $dt --steps 50 --warmup 5 --layout-flags $fl --no-cpu-baseline --rows-file /tmp/rows.json 2>&1 | tail -1 > $OUT/bench_${wl}_${dt}_flags$fl.json
    python3 -c "
import json; d=json.loads(open('$OUT/bench_${wl}_${dt}_flags$fl.json').read()); r=d['roofline']; print('$wl $dt flags $fl', round(d['ms_per_step']*1e3,2), 'us/cycle  min', round(d['timing']['ms_per_step_min']*1e3,2), 'frac', round(r['frac'],4), 'stored', round(r.get('frac_of_stored_bytes',0),4), 'stored bytes', r.get('stored_bytes_per_launch'), 'launches', r.get('launches_per_cycle'))" 2>&1 | tail -1
  done
done | tee $OUT/ab.txt
echo "== parity (full size + cases) of the rows the merge changes"
( timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "secp or peav or arity or nary or bin or unary or mixed" 2>&1 | tail -4 ) | tee $OUT/parity.txt
echo "== serial kernel traces"
cd /tmp
for spec in secp_100k:f64 secp_100k_m4:f64 peav_50k:f64; do
  IFS=: read wl dt <<< "$spec"
  rm -rf $OUT/p
  MAXSUM_NARY_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python3 $R/bench.py --workload $wl --dtype $dt --steps 100 --warmup 10 --no-cpu-baseline --rows-file /tmp/rows.json > $OUT/prof_${wl}_$dt.log 2>&1
  f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_serial_${wl}_${dt}.csv && cut -c1-160 $OUT/kernel_stats_serial_${wl}_${dt}.csv | sed -n 2,8p
  rm -rf $OUT/p
done
exit 0
