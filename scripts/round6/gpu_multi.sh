#!/bin/bash
# Round 6: the workgroup-per-factor kernel in passes (k_factor_nary<.., MULTI>: tables beyond 1 024 entries per value of the first
# variable, arity 6).  BEFORE = layout_flags 33554432 (those factors on factor_generic, a thread per edge); parity of the new
# cases; serial kernel traces.
TAG=${1:-r6_multi}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== parity (new cases first)"
( timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "multi_ or secp_30k_m5 or meeting_5k_d40 or corner" 2>&1 | tail -4 ) | tee $OUT/parity.txt
for spec in secp_30k_m5:f64 secp_30k_m5:f32 meeting_5k_d40:f64 meeting_5k_d40:f32; do
  IFS=: read wl dt <<< "$spec"
  for fl in 33554432 0; do
    st=50; [ $fl != 0 ] && st=5
    timeout 600 python3 bench.py --workload $wl --dtype $dt --steps $st --warmup 2 --layout-flags $fl --no-cpu-baseline --rows-file $OUT/rows_${wl}_${dt}_$fl.json 2>&1 | tail -1 > $OUT/bench_${wl}_${dt}_flags$fl.json
    python3 -c "
import json; d=json.loads(open('$OUT/bench_${wl}_${dt}_flags$fl.json').read()); r=d['roofline']; print('$wl $dt flags $fl', round(d['ms_per_step']*1e3,2), 'us/cycle frac', round(r['frac'],4), 'stored', round(r.get('frac_of_stored_bytes',0),4), 'alg bytes', r.get('algorithmic_bytes_per_launch'), 'launches', r.get('launches_per_cycle'), json.load(open('$OUT/rows_${wl}_${dt}_$fl.json'))['rows'][0].get('factor_kernels'))" 2>&1 | tail -1
  done
done | tee $OUT/ab.txt
echo "== rocprofv3 kernel traces (f64, serial launches)"
cd /tmp
for wl in secp_30k_m5 meeting_5k_d40; do
  rm -rf $OUT/p
  MAXSUM_NARY_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python3 $R/bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --rows-file /tmp/rows.json > $OUT/prof_$wl.log 2>&1
  f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_serial_${wl}_f64.csv && cut -c1-200 $OUT/kernel_stats_serial_${wl}_f64.csv | head -9
  rm -rf $OUT/p
done
exit 0
