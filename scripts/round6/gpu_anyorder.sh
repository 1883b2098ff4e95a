#!/bin/bash
# Round 6: the launches of a cycle after its first WITHOUT the AQL barrier bit (hipExtAnyOrderLaunch, kernels.h LaunchChain):
# cycle times of every multi-launch row with $MAXSUM_ANYORDER = 0 / 1 / 2, the parity tests of those rows under the knob.
TAG=${1:-r6_anyorder}; MODES=${2:-"0 1 2"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for spec in secp_100k:f64 secp_100k:f32 secp_100k_m4:f64 secp_100k_m4:f32 peav_50k:f64 peav_50k:f32 meeting_50k:f64 meeting_50k:f32 coloring_100k_d8:f64 meeting_50k_float:f64 coloring_100k:f64; do
  IFS=: read wl dt <<< "$spec"
  for m in $MODES; do
    MAXSUM_ANYORDER=$m timeout 300 python3 bench.py --workload $wl --dtype $dt --steps 50 --warmup 5 --no-cpu-baseline --rows-file /tmp/rows.json 2>&1 | tail -1 > $OUT/bench_${wl}_${dt}_ao$m.json
    python3 -c "
import json; d=json.loads(open('$OUT/bench_${wl}_${dt}_ao$m.json').read()); r=d['roofline']; print('$wl $dt anyorder $m', round(d['ms_per_step']*1e3,2), 'us/cycle  min', round(d['timing']['ms_per_step_min']*1e3,2), 'frac', round(r['frac'],4), 'launches', r.get('launches_per_cycle'))" 2>&1 | tail -1
  done
done | tee $OUT/ab.txt
echo "== parity under MAXSUM_ANYORDER=1"
( MAXSUM_ANYORDER=1 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4 ) | tee $OUT/parity_ao1.txt
exit 0
