#!/bin/bash
# Round 6: k_factor_small merges its trailing minima through LDS (was: a DPP butterfly, three VALU instructions per 64-bit value and
# step): cycle times of the SECP rows, parity, serial kernel traces.
TAG=${1:-r6_small_lds}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for spec in secp_100k:f64 secp_100k:f32 secp_100k_m4:f64 secp_100k_m4:f32; do
  IFS=: read wl dt <<< "$spec"
  timeout 300 python3 bench.py --workload $wl --dtype $dt --steps 50 --warmup 5 --no-cpu-baseline --rows-file /tmp/rows.json 2>&1 | tail -1 > $OUT/bench_${wl}_${dt}.json
  python3 -c "
import json; d=json.loads(open('$OUT/bench_${wl}_${dt}.json').read()); r=d['roofline']; print('$wl $dt', round(d['ms_per_step']*1e3,2), 'us/cycle  min', round(d['timing']['ms_per_step_min']*1e3,2), 'frac', round(r['frac'],4), 'launches', r.get('launches_per_cycle'))" 2>&1 | tail -1
done | tee $OUT/ab.txt
echo "== parity"
( timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "secp or arity or nary or small or fuzz_maxsum or mixed" 2>&1 | tail -4 ) | tee $OUT/parity.txt
echo "== serial kernel traces"
cd /tmp
for spec in secp_100k:f64 secp_100k_m4:f64 secp_100k_m4:f32; do
  IFS=: read wl dt <<< "$spec"
  rm -rf $OUT/p
  MAXSUM_NARY_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python3 $R/bench.py --workload $wl --dtype $dt --steps 100 --warmup 10 --no-cpu-baseline --rows-file /tmp/rows.json > $OUT/prof_${wl}_$dt.log 2>&1
  f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_serial_${wl}_${dt}.csv && cut -c1-160 $OUT/kernel_stats_serial_${wl}_${dt}.csv | sed -n 2,8p
  rm -rf $OUT/p
done
exit 0
