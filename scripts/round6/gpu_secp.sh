#!/bin/bash
# Round 6: SECP-shaped workloads (generators.secp_like == the reference's `generate secp`): parity of the new cases, cycle times,
# kernel traces.  BEFORE = flag 16 off/on variants are not needed: the arity-5 factors ran on factor_generic until this commit
# (layout_flags 16 = no workgroup-per-factor kernel reproduces it for every n-ary factor).
TAG=${1:-r6_secp}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== parity"
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "(secp or arity5) and test_bit_exact" ) 2>&1 | tail -3
for wl in secp_100k secp_100k_m4; do
  for dt in f64 f32; do
    for fl in 0 16; do
      timeout 300 python3 bench.py --workload $wl --dtype $dt --steps 50 --warmup 5 --layout-flags $fl --no-cpu-baseline --rows-file $OUT/rows_${wl}_${dt}_$fl.json 2>&1 | tail -1 > $OUT/bench_${wl}_${dt}_flags$fl.json
      python3 -c "
import json; d=json.loads(open('$OUT/bench_${wl}_${dt}_flags$fl.json').read()); r=d['roofline']; print('$wl $dt flags $fl', round(d['ms_per_step']*1e3,2), 'us/cycle frac', round(r['frac'],4), 'stored', round(r.get('frac_of_stored_bytes',0),4), r.get('launches_per_cycle'), json.load(open('$OUT/rows_${wl}_${dt}_$fl.json'))['rows'][0]['factor_kernels'])"
    done
  done
done
echo "== rocprofv3 kernel traces (f64)"
cd /tmp
for wl in secp_100k secp_100k_m4; do
  rm -rf $OUT/p
  MAXSUM_NARY_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python3 $R/bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --rows-file /tmp/rows.json > $OUT/prof_$wl.log 2>&1
  f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_serial_${wl}_f64.csv && cut -c1-230 $OUT/kernel_stats_serial_${wl}_f64.csv | head -12
  rm -rf $OUT/p
done
exit 0
