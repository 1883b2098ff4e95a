#!/bin/bash
# Round 6: which launch hosts the K_V_PACK8 variable class -- the group with the most table entries (small-domain groups of arity
# 3 / 4 included: default) against the round-5 rule ($MAXSUM_PACK8_HOST=count: the lane-grid group of the most factors).
TAG=${1:-r6_host}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== parity"
( timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "secp or small or arity or coloring8 or coloring7 or coloring6 or d8 or mixed" 2>&1 | tail -4 ) | tee $OUT/parity.txt
for spec in secp_100k:f64 secp_100k:f32 secp_100k_m4:f64 secp_100k_m4:f32 coloring_100k_d8:f64 coloring_100k_d8:f32; do
  IFS=: read wl dt <<< "$spec"
  for host in count entries; do
    MAXSUM_PACK8_HOST=$host timeout 300 python3 bench.py --workload $wl --dtype $dt --steps 50 --warmup 5 --no-cpu-baseline --rows-file /tmp/rows.json 2>&1 | tail -1 > $OUT/b.json
    python3 -c "
import json; d=json.loads(open('$OUT/b.json').read()); r=d['roofline']; print('$wl $dt host by %-8s' % '$host', round(d['ms_per_step']*1e3,2), 'us/cycle  min', round(d['timing']['ms_per_step_min']*1e3,2), 'frac', round(r['frac'],4), 'launches', r.get('launches_per_cycle'))" 2>&1 | tail -1
  done
done | tee $OUT/ab.txt
echo "== serial kernel traces"
cd /tmp
for spec in secp_100k:f64 secp_100k_m4:f64; do
  IFS=: read wl dt <<< "$spec"
  rm -rf $OUT/p
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python3 $R/bench.py --workload $wl --dtype $dt --steps 100 --warmup 10 --no-cpu-baseline --rows-file /tmp/rows.json > $OUT/prof_${wl}_$dt.log 2>&1
  f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_serial_${wl}_${dt}.csv && cut -c1-170 $OUT/kernel_stats_serial_${wl}_${dt}.csv | sed -n 2,7p
  rm -rf $OUT/p
done
exit 0
