#!/bin/bash
# Round 6: how the wide variable launch shares the machine with the factor launches beside it: the side stream's priority
# ($MAXSUM_SIDE_PRIO = hi / lo) and which side is enqueued first ($MAXSUM_WIDE_LAST=1), on the rows that fork (>= 400 MB per cycle).
TAG=${1:-r6_side}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for spec in peav_50k:f64 peav_50k:f32 meeting_50k:f64 meeting_50k_hetero:f64; do
  IFS=: read wl dt <<< "$spec"
  for cfg in "default::" "prio_hi:h:" "prio_lo:l:" "wide_last::1" "wide_last_hi:h:1" "wide_last_lo:l:1" "one_stream::"; do
    IFS=: read name pr wl_last <<< "$cfg"
    ov=""; [ $name = one_stream ] && ov=0
    MAXSUM_NARY_OVERLAP=$ov MAXSUM_SIDE_PRIO=$pr MAXSUM_WIDE_LAST=$wl_last timeout 300 python3 bench.py --workload $wl --dtype $dt --steps 50 --warmup 5 --no-cpu-baseline --rows-file /tmp/rows.json 2>&1 | tail -1 > $OUT/b.json
    python3 -c "
import json; d=json.loads(open('$OUT/b.json').read()); print('$wl $dt %-14s' % '$name', round(d['ms_per_step']*1e3,2), 'us/cycle  min', round(d['timing']['ms_per_step_min']*1e3,2))" 2>&1 | tail -1
  done
done | tee $OUT/ab.txt
exit 0
